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
    unsigned a_bytes, a2_bytes, b_bytes;   // FAST path: buffer sizes of the stored operands
    int fast;
    size_t slab_stride;    // split-K: floats between partial slabs (out_elems, + N when the column sums ride along as a tail)
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

// Branch-free variant: raw buffer loads whose out-of-range lanes carry an out-of-bounds offset and return 0 (no scalar
// fallback path, no waits inside branches -- the generic loader above compiles to a chain of "load; s_waitcnt vmcnt(0)" blocks).
// Legal when every float4 is 16-byte aligned and wholly inside or outside the matrix (launcher: FAST).
typedef unsigned int u32x4g __attribute__((ext_vector_type(4)));
constexpr unsigned GOOB = 0x7FFFFFF0u;
// VEC == false: the same unit as four dword loads, each with its own range test -- operands whose rows are not 16-byte aligned
// (leading dimension or extent not a multiple of 4: the state-space critics' Linear on [features | latent | 10 labels], K = 4618)
// stay on the branch-free path instead of the generic loader.
// U = 16-byte units per thread and step: 2 with 256 threads, 1 with 512 (the eight-wave tile)
template <bool KCONTIG, bool VEC, int U = 2>
__device__ __forceinline__ void load_step_tile_fast(__amdgpu_buffer_rsrc_t rs, int ld, int R, int r0, int k0, int kend,
                                                    float4 (&reg)[U]) {
    const int tid = threadIdx.x;
    constexpr int NT_ = 512 / U;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        int row, k;
        unsigned off;
        if (KCONTIG) {
            const int unit = tid + NT_ * u;
            row = r0 + (unit >> 3); k = k0 + (unit & 7) * 4;
            off = (unsigned)(row * ld + k) * 4u;
        } else {
            row = r0 + (tid & 15) * 4; k = k0 + u * (NT_ / 16) + (tid >> 4);
            off = (unsigned)(k * ld + row) * 4u;
        }
        if (VEC) {
            const u32x4g t = __builtin_amdgcn_raw_buffer_load_b128(rs, (row < R && k < kend) ? off : GOOB, 0, 0);
            reg[u] = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
        } else {
            unsigned v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = KCONTIG ? (row < R && k + j < kend) : (row + j < R && k < kend);
                v[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, ok ? off + 4u * j : GOOB, 0, 0);
            }
            reg[u] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        }
    }
}

template <bool KCONTIG, int U = 2>
__device__ __forceinline__ void store_step_tile(float* T, const float4 (&reg)[U]) {
    const int tid = threadIdx.x;
    constexpr int NT_ = 512 / U;
    if (KCONTIG) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int unit = tid + NT_ * u, row = unit >> 3, kq = unit & 7;
            float* d = T + (kq * 4) * LDPK + row;
            d[0] = reg[u].x; d[LDPK] = reg[u].y; d[2 * LDPK] = reg[u].z; d[3 * LDPK] = reg[u].w;
        }
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = tid >> 4, r = (tid & 15) * 4;
            *reinterpret_cast<float4*>(T + (u * (NT_ / 16) + kk) * LDP + r) = reg[u];
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
constexpr int TSZ = KSTEP * LDP;                           // one LDS step tile (sized for the wider row)

// NWAVE = 8 (round 4): two waves per SIMD -- waves 4..7 take the second half of every 32-k step (in-workgroup split of the reduction,
// combined through LDS in the epilogue), each thread stages one 16-byte unit per operand and step instead of two.  For the products whose
// grid leaves a CU with ONE workgroup (the critic tail's data gradient: 144 workgroups walking 16 serial steps), where a step is
// bound by its own LDS / L2 round trips with nothing else on the SIMD to issue meanwhile.  FAST operand paths only.
template <bool TA, bool TB, int MASK, int FAST, int NWAVE = 4>
__device__ __forceinline__ void gemm_tile(const GemmParams& P, const int bx, const int by, const int split, float* As, float* Bs) {
    static_assert(NWAVE == 4 || (NWAVE == 8 && FAST != 0), "eight waves: branch-free operand loads only");
    constexpr int U = NWAVE == 4 ? 2 : 1;                  // staging units per thread and step
    constexpr int KG = NWAVE / 4;                          // k-groups of a step
    constexpr int NT_ = 64 * NWAVE;
    // LDS double-buffered so a step costs ONE barrier: the next step's global loads are issued before the MFMA block, parked in
    // registers, and committed to the other buffer after it
    constexpr bool AK = !TA, BKc = TB;                     // operand is k-contiguous in memory
    constexpr int LA = AK ? LDPK : LDP, LB = BKc ? LDPK : LDP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wsub = wave & 3, kg = wave >> 2;
    const int wm = wsub & 1, wn = wsub >> 1, half = lane >> 5, l31 = lane & 31;
    const int m0 = by * BM, n0 = bx * BN;
    const int kb = split * P.kps, ke = min(kb + P.kps, P.K);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool do_colsum = P.colsum != nullptr && by == 0 && tid < BN;   // rows beyond K are zero in the tile
    float csum = 0.f;
    // register ring two steps deep: the loads of step k+2 are issued before the MFMA block of step k, the loads of step k+1
    // (issued one iteration earlier) are committed to the other LDS buffer after it.  A workgroup's step was bound by the
    // latency of ONE prefetch (~1.3 us per 32-k step measured against 0.43 us of MFMA issue); two in flight cover it.
    float4 ra[2][U], rb[2][U], rma[2][U], rmb[2][U];
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)P.A, (short)0, (int)P.a_bytes, 0x00020000);
    const auto rsA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(P.A2 ? P.A2 : P.A), (short)0, (int)(P.A2 ? P.a2_bytes : P.a_bytes), 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)P.B, (short)0, (int)P.b_bytes, 0x00020000);
    const auto rsAr = __builtin_amdgcn_make_buffer_rsrc((void*)(MASK == 1 ? P.a_ref : P.A), (short)0, (int)P.a_bytes, 0x00020000);
    const auto rsBr = __builtin_amdgcn_make_buffer_rsrc((void*)(MASK == 2 ? P.b_ref : P.B), (short)0, (int)P.b_bytes, 0x00020000);
    auto load_step = [&](int k0, float4 (&xa)[U], float4 (&xb)[U], float4 (&xma)[U], float4 (&xmb)[U]) {
        if constexpr (FAST != 0) {
            constexpr bool V = FAST == 1;
            // (source selection by scalar selects, ONE load sequence: no branches around the loads)
            const bool second = P.A2 != nullptr && (AK ? k0 >= P.a_split : m0 >= P.a_split);
            const auto rs = second ? rsA2 : rsA;
            const int ld = second ? P.lda2 : P.lda;
            int R = P.M, r0 = m0, kk0 = k0, kke = ke;
            if (P.A2 != nullptr) {
                if (AK) { kk0 = second ? k0 - P.a_split : k0; kke = second ? ke - P.a_split : min(ke, P.a_split); }
                else { R = second ? P.M - P.a_split : P.a_split; r0 = second ? m0 - P.a_split : m0; }
            }
            load_step_tile_fast<AK, V, U>(rs, ld, R, r0, kk0, kke, xa);
            if (MASK == 1) load_step_tile_fast<AK, V, U>(rsAr, P.lda, P.M, m0, k0, ke, xma);
            load_step_tile_fast<BKc, V, U>(rsB, P.ldb, P.N, n0, k0, ke, xb);
            if (MASK == 2) load_step_tile_fast<BKc, V, U>(rsBr, P.ldb, P.N, n0, k0, ke, xmb);
            return;
        } else {
        if (P.A2 == nullptr) {
            load_step_tile<AK>(P.A, P.lda, P.M, m0, k0, ke, P.vecA, xa);
        } else if (AK) {        // element (m, k) at A[m*lda + k]: the sources split the k range
            const bool second = k0 >= P.a_split;
            load_step_tile<AK>(second ? P.A2 - P.a_split : P.A, second ? P.lda2 : P.lda, P.M, m0, k0, second ? ke : min(ke, P.a_split),
                               P.vecA, xa);
        } else {                // element (m, k) at A[k*lda + m]: the sources split the m range
            const bool second = m0 >= P.a_split;
            load_step_tile<AK>(second ? P.A2 - P.a_split : P.A, second ? P.lda2 : P.lda, second ? P.M : P.a_split, m0, k0, ke,
                               P.vecA, xa);
        }
        if (MASK == 1) load_step_tile<AK>(P.a_ref, P.lda, P.M, m0, k0, ke, P.vecA, xma);
        load_step_tile<BKc>(P.B, P.ldb, P.N, n0, k0, ke, P.vecB, xb);
        if (MASK == 2) load_step_tile<BKc>(P.b_ref, P.ldb, P.N, n0, k0, ke, P.vecB, xmb);
        }
    };
    auto store_step = [&](int buf, float4 (&xa)[U], float4 (&xb)[U], float4 (&xma)[U], float4 (&xmb)[U]) {
        if (MASK == 1) {
#pragma unroll
            for (int u = 0; u < U; ++u) xa[u] = apply_mask<true>(xa[u], xma[u], P.ref_act, P.ref_alpha);
        }
        if (MASK == 2) {
#pragma unroll
            for (int u = 0; u < U; ++u) xb[u] = apply_mask<true>(xb[u], xmb[u], P.ref_act, P.ref_alpha);
        }
        store_step_tile<AK, U>(As + buf * TSZ, xa);
        store_step_tile<BKc, U>(Bs + buf * TSZ, xb);
    };
    auto mma_step = [&](int buf) {
        const float* Ab = As + buf * TSZ;
        const float* Bb = Bs + buf * TSZ;
        if (do_colsum) {
#pragma unroll
            for (int kk = 0; kk < KSTEP; ++kk) csum += Bb[kk * LB + tid];
        }
        // all 32 fragment reads of the step are issued up front and the 16 MFMAs drain them with counted lgkmcnt waits.  The
        // straightforward loop compiles to "2 ds_read; s_waitcnt lgkmcnt(0); v_mfma" per k-pair: one LDS round trip of matrix-pipe
        // idle per MFMA (measured 1.07 us per step where the MFMAs need 0.43 us) -- nothing else hides it when the grid puts
        // one or two workgroups on a CU, which is the regime of the step's Linear layers.
        constexpr int NM = KSTEP / 2 / KG;                       // MFMAs of this wave per step (its k-group's half of the step with 8 waves)
        float fa[NM], fb[NM];
        const int kbase = kg * (KSTEP / KG);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            fa[i] = Ab[(kbase + 2 * i + half) * LA + wm * 32 + l31];
            fb[i] = Bb[(kbase + 2 * i + half) * LB + wn * 32 + l31];
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 12 / KG, 0); // the first DS reads
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[i], acc, 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA ...
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // ... then the next two reads still to issue
        }
    };
    load_step(kb, ra[0], rb[0], rma[0], rmb[0]);
    store_step(0, ra[0], rb[0], rma[0], rmb[0]);
    if (kb + KSTEP < ke) load_step(kb + KSTEP, ra[1], rb[1], rma[1], rmb[1]);
    __syncthreads();
    // two steps per trip so that the register sets are compile-time: even steps multiply buffer 0 and refill set 0
    for (int k0 = kb; k0 < ke; k0 += 2 * KSTEP) {
        if (k0 + 2 * KSTEP < ke) load_step(k0 + 2 * KSTEP, ra[0], rb[0], rma[0], rmb[0]);
        mma_step(0);
        if (k0 + KSTEP < ke) store_step(1, ra[1], rb[1], rma[1], rmb[1]);
        __syncthreads();
        if (k0 + KSTEP >= ke) break;
        if (k0 + 3 * KSTEP < ke) load_step(k0 + 3 * KSTEP, ra[1], rb[1], rma[1], rmb[1]);
        mma_step(1);
        if (k0 + 2 * KSTEP < ke) store_step(0, ra[0], rb[0], rma[0], rmb[0]);
        __syncthreads();
    }
    if (do_colsum && n0 + tid < P.N) P.colsum[(P.SK > 1 ? (size_t)split * P.slab_stride : 0) + n0 + tid] = csum;   // (split-K: a partial sum in the slab's tail)
    // epilogue: global stores are issue-bound (16 dword stores per lane took longer than the k-loop of the short-K layers), so
    // the tile goes through LDS and leaves as one float4 row segment per thread and pass (4 passes)
    constexpr int LC = BN + 4;
    float* Cs = As;                                   // 64 x 68 floats <= 2 * TSZ (all waves are past the last barrier)
    float* Cs2 = Bs;                                  // eight waves: the second k-group's partial tile, added in the store passes
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ml = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        (kg ? Cs2 : Cs)[ml * LC + wn * 32 + l31] = acc[r];
    }
    __syncthreads();
    const bool direct = P.SK == 1;
    float* Cp = direct ? P.C : P.C + (size_t)split * P.slab_stride;
    int ldc = P.N, ncol0 = 0, nend = P.N;       // output columns [ncol0, nend) live in Cp with leading dimension ldc
    if (P.C2 != nullptr) {
        if (n0 >= P.c_split) { Cp = P.C2; ldc = P.N - P.c_split; ncol0 = P.c_split; }
        else { ldc = P.c_split; nend = P.c_split; }
    }
    const bool vec_out = (ldc & 3) == 0 && ((uintptr_t)Cp & 15) == 0;
#pragma unroll
    for (int pass = 0; pass < 1024 / NT_; ++pass) {
        const int idx = tid + pass * NT_, ml = idx >> 4, c4 = (idx & 15) * 4;
        const int m = m0 + ml, n = n0 + c4;
        if (m >= P.M || n >= P.N) continue;
        float4 v = *reinterpret_cast<const float4*>(Cs + ml * LC + c4);
        if constexpr (KG == 2) {
            const float4 w2 = *reinterpret_cast<const float4*>(Cs2 + ml * LC + c4);
            v.x += w2.x; v.y += w2.y; v.z += w2.z; v.w += w2.w;
        }
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

template <bool TA, bool TB, int MASK = 0, int FAST = 0>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams P) {
    warm_kernarg(P);
    __shared__ __attribute__((aligned(16))) float As[2 * TSZ];
    __shared__ __attribute__((aligned(16))) float Bs[2 * TSZ];
    gemm_tile<TA, TB, MASK, FAST>(P, blockIdx.x, blockIdx.y, blockIdx.z, As, Bs);
}

template <bool TA, bool TB, int MASK = 0, int FAST = 1>
__global__ __launch_bounds__(512) void gemm_kernel8(const GemmParams P) {
    warm_kernarg(P);
    __shared__ __attribute__((aligned(16))) float As[2 * TSZ];
    __shared__ __attribute__((aligned(16))) float Bs[2 * TSZ];
    gemm_tile<TA, TB, MASK, FAST, 8>(P, blockIdx.x, blockIdx.y, blockIdx.z, As, Bs);
}

// Independent products in ONE launch (a Linear layer's weight gradient next to its data gradient: both consume the same
// upstream gradient, neither fills the chip, and every launch of a step graph costs ~5 us before it does any work).
// Workgroups [first[j], first[j+1]) belong to problem j; FAST operands, no split-K.
// The BCE terms of a cost whose logits are the rows of ONE critic head, known before the head runs (the caller's hint: consecutive
// row ranges, label, weight -- tflib/objs/gan_inference.py:104-117).  The cost's gradient for a unit upstream gradient is row-local,
// g[r] = w_k / n_k * (sigmoid(logit[r]) - z_k), so the forward tail can leave g and gh = g w_out^T lrelu'(h) behind at once and the
// head's backward products follow it directly: one launch less on the chain  tail product -> logits -> cost -> head backward -> products
// of every step.  What needs all rows (the cost itself, d_wout = h^T g, d_bout) rides in the products' launch (head_tail_wg).
// kind 0: sigmoid cross-entropy terms (label z); kind 1: plain means (the Wasserstein costs, gan_inference.py:4-45: g[r] = w_k / n_k).
// ext[k] != NULL: term k is not a row range of the logits but `rows[k]` values elsewhere (the one-element gradient penalty of a wali-gp
// critic cost): it only enters the cost's value.
struct HeadTerms { int count, kind; int rows[4]; float z[4]; float w[4]; const float* ext[4]; };

// What the cost and the head's own parameters need from ALL rows, as extra workgroups of the products' launch (or head_tail_k):
// workgroup 0: the cost -- bce_multi_fwd_k's loop and summation order on 256 threads (pointwise.hip);  workgroups 1..: 16 columns of
// d_wout = h^T g each, the first of them d_bout too -- head_out_bwd_k's loop and combination order without the gh store.  Threads
// beyond 256 (eight-wave launches) only keep the barriers company: their partial sums are exact zeros.
struct HeadTail {
    const float* logits; const float* g; const float* h;
    float* d_wout; float* d_bout; float* loss;
    int M, H, first;
    HeadTerms terms;
};

__device__ __forceinline__ void head_tail_wg(const HeadTail& T, const int local, float* lds /* >= 16 * 16 + 32 floats */) {
    const int tid = threadIdx.x;
    const bool on = tid < 256;
    if (local == 0) {
        float* sm = lds;
        float tot = 0.f;
        int r0 = 0;
        for (int k = 0; k < T.terms.count; ++k) {
            const float z = T.terms.z[k];
            const int n = T.terms.rows[k];
            const float* x = T.terms.ext[k] ? T.terms.ext[k] : T.logits + r0;
            float s = 0.f;
            if (on)
                for (int i = tid; i < n; i += 256) {
                    const float v = x[i];
                    s += T.terms.kind == 1 ? v : fmaxf(v, 0.f) - v * z + log1pf(expf(-fabsf(v)));
                }
            s = block_sum(s, sm);
            const float r = T.terms.w[k] * (s / (float)n);
            tot = k ? tot + r : r;
            if (!T.terms.ext[k]) r0 += n;
        }
        if (tid == 0 && T.loss) T.loss[0] = tot;
        return;
    }
    float (*red)[16] = reinterpret_cast<float (*)[16]>(lds);
    const int cl = tid & 15, rg = (tid >> 4) & 15;
    const int c = (local - 1) * 16 + cl;
    float acc = 0.f, gs = 0.f;
    if (on && c < T.H) {
#pragma unroll 8
        for (int r = rg; r < T.M; r += 16) {
            const float gr = T.g[r], hv = T.h[(size_t)r * T.H + c];
            acc = fmaf(gr, hv, acc);
            gs += gr;
        }
    }
    if (on) red[rg][cl] = acc;
    __syncthreads();
    if (on && rg == 0 && c < T.H && T.d_wout)
        T.d_wout[c] = (((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) + ((red[4][cl] + red[5][cl]) + (red[6][cl] + red[7][cl]))) +
                      (((red[8][cl] + red[9][cl]) + (red[10][cl] + red[11][cl])) + ((red[12][cl] + red[13][cl]) + (red[14][cl] + red[15][cl])));
    if (local == 1 && T.d_bout) {
        __syncthreads();
        if (on && cl == 0) red[rg][0] = gs;
        __syncthreads();
        if (tid == 0) T.d_bout[0] = (((red[0][0] + red[1][0]) + (red[2][0] + red[3][0])) + ((red[4][0] + red[5][0]) + (red[6][0] + red[7][0]))) +
                                    (((red[8][0] + red[9][0]) + (red[10][0] + red[11][0])) + ((red[12][0] + red[13][0]) + (red[14][0] + red[15][0])));
    }
}

__global__ __launch_bounds__(256) void head_tail_k(const HeadTail T) {
    __shared__ float lds[16 * 16 + 32];
    head_tail_wg(T, blockIdx.x, lds);
}

struct GemmGroup {
    int n;
    int first[4];
    int kind[3];        // 0: A^T B (ta)   1: A B^T (tb)   2: A B
    int gx[3];
    GemmParams p[3];
    int has_tail;       // extra workgroups behind the products: head_tail_wg (tail.first = their first workgroup)
    HeadTail tail;
};

template <int NWAVE>
__global__ __launch_bounds__(64 * NWAVE) void gemm_group_kernel(const GemmGroup G) {
    warm_kernarg(G);
    __shared__ __attribute__((aligned(16))) float As[2 * TSZ];
    __shared__ __attribute__((aligned(16))) float Bs[2 * TSZ];
    if (G.has_tail && (int)blockIdx.x >= G.tail.first) {
        head_tail_wg(G.tail, (int)blockIdx.x - G.tail.first, As);
        return;
    }
    int j = 0;
    if (G.n > 1 && (int)blockIdx.x >= G.first[1]) j = 1;
    if (G.n > 2 && (int)blockIdx.x >= G.first[2]) j = 2;
    const int local = blockIdx.x - G.first[j];
    const int bx = local % G.gx[j], by = local / G.gx[j];
    const GemmParams& P = G.p[j];
    switch (G.kind[j]) {
        case 0: gemm_tile<true, false, 0, 1, NWAVE>(P, bx, by, 0, As, Bs); break;
        case 1: gemm_tile<false, true, 0, 1, NWAVE>(P, bx, by, 0, As, Bs); break;
        default: gemm_tile<false, false, 0, 1, NWAVE>(P, bx, by, 0, As, Bs); break;
    }
}
static bool group_w8() {
    static const bool v = [] { const char* e = getenv("GGAN_GEMM_GROUP_W8"); return e ? atoi(e) != 0 : true; }();
    return v;
}
#define GGAN_LAUNCH_GROUP(fl, nwg, s, GG)                                                                                        \
    do {                                                                                                                          \
        if (group_w8()) { GGAN_LAUNCH("gemm_group_kernel<8>", fl, 0, gemm_group_kernel<8>, dim3(nwg), dim3(512), 0, s, GG); }    \
        else { GGAN_LAUNCH("gemm_group_kernel", fl, 0, gemm_group_kernel<4>, dim3(nwg), dim3(256), 0, s, GG); }                  \
    } while (0)

// ---- the critics' heads: Linear + LeakyReLU + Linear(H -> 1) -------------------------------------------------------------------
// forward tail: h = lrelu(sum of the split-K slabs of the first Linear + b), logits = h . w_out + b_out.  One workgroup per two rows.
__global__ __launch_bounds__(256) void head_out_fwd_k(const float* __restrict__ part, int SK, size_t slab, const float* __restrict__ b,
                                                      const float* __restrict__ w_out, const float* __restrict__ b_out, float alpha,
                                                      float* __restrict__ h, float* __restrict__ logits, int M, int H) {
    __shared__ float red[4];
    const int tid = threadIdx.x, rl = tid >> 7, t = tid & 127;
    const int row = blockIdx.x * 2 + rl;
    float dot = 0.f;
    if (row < M) {
        for (int c = t * 4; c < H; c += 512) {
            const size_t o = (size_t)row * H + c;
            float4 v = *reinterpret_cast<const float4*>(part + o);
            // (slab order: deterministic.  Unrolled: the slab loads are independent, only the adds are ordered -- rolled, the 16
            //  slabs of the K = 4608 tail were 16 serial L2 round trips, most of this kernel's 8 us)
#pragma unroll 8
            for (int s = 1; s < SK; ++s) {
                const float4 u = *reinterpret_cast<const float4*>(part + (size_t)s * slab + o);
                v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
            }
            const float4 bb = *reinterpret_cast<const float4*>(b + c);
            const float4 ww = *reinterpret_cast<const float4*>(w_out + c);
            v.x = fmaxf(alpha * (v.x + bb.x), v.x + bb.x); v.y = fmaxf(alpha * (v.y + bb.y), v.y + bb.y);
            v.z = fmaxf(alpha * (v.z + bb.z), v.z + bb.z); v.w = fmaxf(alpha * (v.w + bb.w), v.w + bb.w);
            *reinterpret_cast<float4*>(h + o) = v;
            dot = fmaf(v.x, ww.x, fmaf(v.y, ww.y, fmaf(v.z, ww.z, fmaf(v.w, ww.w, dot))));
        }
    }
    dot = wave_sum(dot);
    if ((tid & 63) == 0) red[tid >> 6] = dot;
    __syncthreads();
    if (t == 0 && row < M) logits[row] = (red[2 * rl] + red[2 * rl + 1]) + b_out[0];
}

// head_out_fwd_k + g[r] + gh[r,:] (H <= 2048: the row's h values stay in registers).  Same expressions as bce_multi_fwd_k / head_out_bwd_k.
__global__ __launch_bounds__(256) void head_out_fwd_bce_k(const float* __restrict__ part, int SK, size_t slab, const float* __restrict__ b,
                                                          const float* __restrict__ w_out, const float* __restrict__ b_out, float alpha,
                                                          float* __restrict__ h, float* __restrict__ logits, int M, int H,
                                                          const HeadTerms T, float* __restrict__ g_out, float* __restrict__ gh) {
    __shared__ float red[4];
    const int tid = threadIdx.x, rl = tid >> 7, t = tid & 127;
    const int row = blockIdx.x * 2 + rl;
    float dot = 0.f;
    float4 hv[4], wv[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = t * 4 + it * 512;
        if (row < M && c < H) {
            const size_t o = (size_t)row * H + c;
            float4 v = *reinterpret_cast<const float4*>(part + o);
#pragma unroll 8
            for (int s = 1; s < SK; ++s) {
                const float4 u = *reinterpret_cast<const float4*>(part + (size_t)s * slab + o);
                v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
            }
            const float4 bb = *reinterpret_cast<const float4*>(b + c);
            const float4 ww = *reinterpret_cast<const float4*>(w_out + c);
            v.x = fmaxf(alpha * (v.x + bb.x), v.x + bb.x); v.y = fmaxf(alpha * (v.y + bb.y), v.y + bb.y);
            v.z = fmaxf(alpha * (v.z + bb.z), v.z + bb.z); v.w = fmaxf(alpha * (v.w + bb.w), v.w + bb.w);
            *reinterpret_cast<float4*>(h + o) = v;
            dot = fmaf(v.x, ww.x, fmaf(v.y, ww.y, fmaf(v.z, ww.z, fmaf(v.w, ww.w, dot))));
            hv[it] = v; wv[it] = ww;
        }
    }
    dot = wave_sum(dot);
    if ((tid & 63) == 0) red[tid >> 6] = dot;
    __syncthreads();
    if (row >= M) return;
    const float lg = (red[2 * rl] + red[2 * rl + 1]) + b_out[0];
    if (t == 0) logits[row] = lg;
    float z = 0.f, sc = 0.f;
    {
        int r0 = 0;
        for (int k = 0; k < T.count; ++k) {
            if (T.ext[k]) continue;
            if (row >= r0 && row < r0 + T.rows[k]) { z = T.z[k]; sc = 1.f * T.w[k] / (float)T.rows[k]; }
            r0 += T.rows[k];
        }
    }
    const float gr = T.kind == 1 ? sc : sc * (1.f / (1.f + expf(-lg)) - z);
    if (t == 0) g_out[row] = gr;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = t * 4 + it * 512;
        if (c < H) {
            float4 o4;
            o4.x = gr * wv[it].x * (hv[it].x > 0.f ? 1.f : alpha); o4.y = gr * wv[it].y * (hv[it].y > 0.f ? 1.f : alpha);
            o4.z = gr * wv[it].z * (hv[it].z > 0.f ? 1.f : alpha); o4.w = gr * wv[it].w * (hv[it].w > 0.f ? 1.f : alpha);
            *reinterpret_cast<float4*>(gh + (size_t)row * H + c) = o4;
        }
    }
}

// backward head: gh[r,c] = g[r] * w_out[c] * lrelu'(h[r,c]);  d_wout[c] = sum_r g[r] h[r,c];  d_bout = sum_r g[r].
// One workgroup per 16 columns, 16 row groups (measured against 64 x 4 and 32 x 8: the launch is a chain of L2 round trips per
// thread, not bandwidth -- 8 rows per thread at 128 rows, all of their loads in flight at once); the row groups are combined in a
// fixed order (deterministic).  bce_head_bwd_k (pointwise.hip) is the same loop with g
// formed from the logits.
__global__ __launch_bounds__(256) void head_out_bwd_k(const float* __restrict__ g, const float* __restrict__ h,
                                                      const float* __restrict__ w_out, float alpha, float* __restrict__ gh,
                                                      float* __restrict__ d_wout, float* __restrict__ d_bout, int M, int H) {
    __shared__ float red[16][16];
    const int tid = threadIdx.x, cl = tid & 15, rg = tid >> 4;
    const int c = blockIdx.x * 16 + cl;
    float acc = 0.f, gs = 0.f;
    if (c < H) {
        const float w = w_out[c];
#pragma unroll 8
        for (int r = rg; r < M; r += 16) {
            const float gr = g[r], hv = h[(size_t)r * H + c];
            gh[(size_t)r * H + c] = gr * w * (hv > 0.f ? 1.f : alpha);
            acc = fmaf(gr, hv, acc);
            gs += gr;
        }
    }
    red[rg][cl] = acc;
    __syncthreads();
    if (rg == 0 && c < H && d_wout)
        d_wout[c] = (((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) + ((red[4][cl] + red[5][cl]) + (red[6][cl] + red[7][cl]))) +
                    (((red[8][cl] + red[9][cl]) + (red[10][cl] + red[11][cl])) + ((red[12][cl] + red[13][cl]) + (red[14][cl] + red[15][cl])));
    if (blockIdx.x == 0 && d_bout) {
        __syncthreads();
        if (cl == 0) red[rg][0] = gs;
        __syncthreads();
        if (tid == 0) d_bout[0] = (((red[0][0] + red[1][0]) + (red[2][0] + red[3][0])) + ((red[4][0] + red[5][0]) + (red[6][0] + red[7][0]))) +
                                (((red[8][0] + red[9][0]) + (red[10][0] + red[11][0])) + ((red[12][0] + red[13][0]) + (red[14][0] + red[15][0])));
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

// ---- skinny products: at most a few hundred 16x16 tiles of C, K <= 1024 -- ONE launch, no split-K slabs ------------------------------
// The Linear layers of the code-space critics (128 rows x 512 x 512, gmgan_inference_cifar10.py:255-291) and of the z paths fill 16 of
// the 64x64 tiles above, so they were spread over the chip by split-K: a product launch walking 2 k-steps per workgroup plus a reduce
// launch (bias + activation), 9 + 5 us for 0.07 GFLOP, and ten such pairs per gmgan iteration.  Here a workgroup owns ONE 16x16 tile
// of C and its eight waves split K between them (16-k steps dealt round-robin).  No LDS staging: a wave fetches its operand
// fragments straight into the v_mfma_f32_16x16x4_f32 register layout -- lane (row l & 15, k slot l >> 4) holds 4 consecutive k of its
// row (one 16-byte load where the operand is k-contiguous and aligned, else four dword loads whose 16 lanes per k row form a 64-byte
// run), so MFMA j of a step multiplies the k values {k0 + 4 slot + j}: any k order is a legal summation order as long as both operands
// use the same one.  ALL loads of a wave are in flight at once (NS steps, compile time), one L2 round trip per product; the eight partial
// tiles are added through LDS in wave order (deterministic), + bias + activation, 64-byte store runs.
struct SkinnyParams {
    const float* A; const float* B; const float* bias; float* C; const float* a_ref;
    const float* A2;       // optional: columns k >= a_split (a multiple of 16: a 16-k step has ONE source) of A come from A2 [M, K - a_split]
    float* C2;             // optional: output columns >= c_split (a multiple of 16) go to C2 [M, N - c_split], the others to C [M, c_split]
    int a_split, c_split, vecA2;
    int M, N, K, lda, ldb, act, ref_act, vecA, vecB;
    float alpha, ref_alpha;
    unsigned a_bytes, b_bytes, a2_bytes;
};

template <bool TB, bool MASK, int NS>
__global__ __launch_bounds__(512) void gemm_skinny_k(const SkinnyParams P) {
    __shared__ float red[8 * 4 * 64];
    constexpr unsigned SOOB = 0x7FFFFFF0u;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = blockIdx.y * 16 + l15, n = blockIdx.x * 16 + l15;
    const auto ra = __builtin_amdgcn_make_buffer_rsrc((void*)P.A, (short)0, (int)P.a_bytes, 0x00020000);
    const auto rr = __builtin_amdgcn_make_buffer_rsrc((void*)(MASK ? P.a_ref : P.A), (short)0, (int)P.a_bytes, 0x00020000);
    const auto rb = __builtin_amdgcn_make_buffer_rsrc((void*)P.B, (short)0, (int)P.b_bytes, 0x00020000);
    const auto ra2 = __builtin_amdgcn_make_buffer_rsrc((void*)(P.A2 ? P.A2 : P.A), (short)0, (int)(P.A2 ? P.a2_bytes : P.a_bytes), 0x00020000);
    const int KA = P.A2 ? P.a_split : P.K;               // k extent of the first source
    float a[NS][4], b[NS][4], r[MASK ? NS : 1][4];
    // k-contiguous operand: row `row` (valid when row_ok), 4 consecutive k from kk of the kend the source holds
    auto load_kc = [&](decltype(ra) rs, bool vec, int row, bool row_ok, int ld, int kk, int kend, float* dst) {
        const unsigned base = (unsigned)(row * ld + kk) * 4u;
        if (vec) {
            const u32x4g t = __builtin_amdgcn_raw_buffer_load_b128(rs, (row_ok && kk < kend) ? base : SOOB, 0, 0);
            dst[0] = __uint_as_float(t.x); dst[1] = __uint_as_float(t.y); dst[2] = __uint_as_float(t.z); dst[3] = __uint_as_float(t.w);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                dst[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (row_ok && kk + j < kend) ? base + 4u * j : SOOB, 0, 0));
        }
    };
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int kk = (wave + 8 * i) * 16 + 4 * kq;
        if (P.A2 && (wave + 8 * i) * 16 >= P.a_split) load_kc(ra2, P.vecA2 != 0, m, m < P.M, P.K - P.a_split, kk - P.a_split, P.K - P.a_split, a[i]);
        else load_kc(ra, P.vecA != 0, m, m < P.M, P.lda, kk, KA, a[i]);
        if (MASK) load_kc(rr, P.vecA != 0, m, m < P.M, P.lda, kk, P.K, r[i]);
        if (TB) load_kc(rb, P.vecB != 0, n, n < P.N, P.ldb, kk, P.K, b[i]);
        else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                b[i][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb, (n < P.N && kk + j < P.K) ? (unsigned)((kk + j) * P.ldb + n) * 4u : SOOB, 0, 0));
        }
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float av = MASK ? act_grad(a[i][j], r[i][j], P.ref_act, P.ref_alpha) : a[i][j];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[i][j], acc, 0, 0, 0);
        }
#pragma unroll
    for (int q = 0; q < 4; ++q) red[(wave * 4 + q) * 64 + lane] = acc[q];
    __syncthreads();
    if (tid < 256) {
        // accumulator register q of lane l is C[4 (l >> 4) + q][l & 15]
        const int q = tid >> 6;
        float v = red[q * 64 + lane];
#pragma unroll
        for (int w = 1; w < 8; ++w) v += red[(w * 4 + q) * 64 + lane];
        const int row = blockIdx.y * 16 + 4 * kq + q;
        if (row < P.M && n < P.N) {
            if (P.bias) v += P.bias[n];
            v = act_apply(v, P.act, P.alpha);
            if (P.C2 && n >= P.c_split) P.C2[(size_t)row * (P.N - P.c_split) + (n - P.c_split)] = v;
            else P.C[(size_t)row * (P.C2 ? P.c_split : P.N) + n] = v;
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

// Fill the parameter block of one product (operand views, float4 legality, split-K choice, FAST path); no launch.
// mode 0: normal (split-K result reduced by the caller)   1: no split-K (grouped launches, fused column sums)
constexpr int kSplitTarget = 256;     // workgroups a split-K product aims at (the critic head's forward asks for more: split_target)

static int head_split_target() {
    static const int v = [] { const char* e = getenv("GGAN_HEAD_WGS"); return e ? atoi(e) : 512; }();
    return v;
}

struct GemmPlan {
    GemmParams P;
    int gx, gy;
    void* ws;
};

static int gemm_plan(GemmPlan& G, int mode, int ta, int tb, int M, int N, int K, const float* A, const float* B, const float* bias,
                     float* C, float* colsum, int act, float alpha, void* ws, size_t ws_bytes, const float* a_ref = nullptr,
                     const float* b_ref = nullptr, int ref_act = 0, float ref_alpha = 0.f, const float* A2 = nullptr,
                     int a_split = 0, float* C2 = nullptr, int c_split = 0, int split_target = kSplitTarget) {
    GemmParams& P = G.P;
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
    G.ws = ws;
    const int gx = cdiv(N, BN), gy = cdiv(M, BM);
    G.gx = gx; G.gy = gy;
    int sk = 1;
    if (mode == 0 && ws && !C2) {       // (a split output needs the whole K range in one workgroup; column sums ride along as slab tails)
        const char* e = getenv("GGAN_GEMM_SK");
        if (e) sk = atoi(e);
        else {
            // a workgroup's k-step is latency-bound (~0.6 us measured), so spread K over the idle CUs down to one
            // BK-step per split; measured optimum on the hot path's shapes (tools/bench_gemm.py)
            const int base = gx * gy;
            // (long reductions -- the Conv3D patch-matrix filter gradients, K = 10^4..10^6 rows: four workgroups per CU and up to
            // 256 slabs, so each workgroup's serial chain stays ~10^3 rows)
            const bool tall = K >= 16384;
            sk = (tall ? 1024 : split_target) / base;
            const int max_sk = K / BK;
            if (sk > max_sk) sk = max_sk;
            if (sk > (tall ? 256 : split_target / 4)) sk = tall ? 256 : split_target / 4;
            // short reductions over a grid that already covers a good part of the chip (the weight gradients of the batch-64
            // layers): the extra reduce launch costs more than it spreads (5.4 vs 8.4 us measured at 512x512x64)
            if (K <= 128 && base >= 32) sk = 1;
        }
        if (sk < 1) sk = 1;
        // with column sums the unsplit product is one launch and the split one two: split only where the serial chain is long
        // and the grid small (the weight gradients of the 1024-row critics: 8..24 workgroups walking 32 steps, 40 us)
        if (colsum && !(gx * gy < 64 && K >= 512)) sk = 1;
        while (sk > 1 && (size_t)sk * (P.out_elems + (colsum ? N : 0)) * sizeof(float) > ws_bytes) sk /= 2;
    }
    // (a two-source operand is switched per k-STEP: every split must start on the step grid, or a step would straddle a_split)
    const int kq = (A2 && !ta) ? KSTEP : BK;
    P.kps = cdiv(cdiv(K, sk), kq) * kq;
    P.SK = cdiv(K, P.kps);
    P.slab_stride = P.out_elems + ((colsum && P.SK > 1) ? (size_t)N : 0);
    if (P.SK > 1) {
        P.C = (float*)ws;
        if (colsum) P.colsum = (float*)ws + P.out_elems;      // slab s: [out_elems product | N column sums]
    }
    // branch-free buffer-load path: every operand float4 aligned and wholly in or out of range, byte offsets within 31 bits
    const size_t szA = (size_t)(A2 ? (ta ? (size_t)K * a_split : (size_t)M * a_split) : (size_t)M * K) * 4;
    const size_t szA2 = A2 ? (size_t)(ta ? (size_t)K * (M - a_split) : (size_t)M * (K - a_split)) * 4 : 0;
    const size_t szB = (size_t)N * K * 4;
    const bool dims4 = (K % 4 == 0) && (P.kps % 4 == 0) && (ta ? (M % 4 == 0 && (!A2 || a_split % 4 == 0)) : true) && (tb ? true : N % 4 == 0);
    const bool small = szA < 0x7FFFFFF0ull && szA2 < 0x7FFFFFF0ull && szB < 0x7FFFFFF0ull && !getenv("GGAN_GEMM_GENERIC");
    P.fast = !small ? 0 : ((P.vecA && P.vecB && dims4) ? 1 : 2);        // 1: 16-byte loads, 2: dword loads (any alignment / extent)
    P.a_bytes = (unsigned)szA; P.a2_bytes = (unsigned)szA2; P.b_bytes = (unsigned)szB;
    return 0;
}

// launch one planned product (split-K slabs stay in the workspace: the caller reduces or consumes them)
static int gemm_launch_planned(const GemmPlan& G, int ta, int tb, hipStream_t s) {
    const GemmParams& P = G.P;
    const double fl = 2.0 * P.M * P.N * (double)P.K;
    const dim3 grid(G.gx, G.gy, P.SK), block(256);
    // eight waves (in-workgroup split of every k-step) where the grid leaves at most ~one workgroup per CU: nothing else on the SIMD
    // hides a step's LDS / L2 round trips there.  GGAN_GEMM_W8 = 0: never, N: grids up to N workgroups (default 256)
    static const int w8_max = [] { const char* e = getenv("GGAN_GEMM_W8"); return e ? atoi(e) : (1 << 30); }();
    static const bool w8_colsum = [] { const char* e = getenv("GGAN_GEMM_W8_COLSUM"); return e ? atoi(e) != 0 : true; }();
    static const int w8_min_steps = [] { const char* e = getenv("GGAN_GEMM_W8_MIN_STEPS"); return e ? atoi(e) : 2; }();
    // (a workgroup that walks ONE step has no round trip between steps to hide: the split would only add its combine)
    const bool w8 = P.fast != 0 && (long)G.gx * G.gy * P.SK <= (long)w8_max && (!P.colsum || w8_colsum) && P.kps >= w8_min_steps * KSTEP;
#define GGAN_GEMM_CASE(TA_, TB_, MK_, NAME_)                                                                                      \
    do {                                                                                                                           \
        if (w8 && P.fast == 1) { GGAN_LAUNCH("gemm_kernel8" NAME_, fl, 0, (gemm_kernel8<TA_, TB_, MK_, 1>), grid, dim3(512), 0, s, P); }   \
        else if (w8) { GGAN_LAUNCH("gemm_kernel8" NAME_, fl, 0, (gemm_kernel8<TA_, TB_, MK_, 2>), grid, dim3(512), 0, s, P); }            \
        else if (P.fast == 1) { GGAN_LAUNCH("gemm_kernel" NAME_, fl, 0, (gemm_kernel<TA_, TB_, MK_, 1>), grid, block, 0, s, P); }              \
        else if (P.fast == 2) { GGAN_LAUNCH("gemm_kernel" NAME_, fl, 0, (gemm_kernel<TA_, TB_, MK_, 2>), grid, block, 0, s, P); }               \
        else { GGAN_LAUNCH("gemm_kernel" NAME_, fl, 0, (gemm_kernel<TA_, TB_, MK_, 0>), grid, block, 0, s, P); }                                \
    } while (0)
    if (P.a_ref || P.b_ref) {
        if (P.a_ref && !P.b_ref && !ta && tb) GGAN_GEMM_CASE(false, true, 1, "<false, true, 1>");
        else if (P.b_ref && !P.a_ref && ta && !tb) GGAN_GEMM_CASE(true, false, 2, "<true, false, 2>");
        else { set_error("gemm: unsupported mask/transposition combination"); return -1; }
    } else
    if (!ta && !tb) GGAN_GEMM_CASE(false, false, 0, "<false, false>");
    else if (!ta && tb) GGAN_GEMM_CASE(false, true, 0, "<false, true>");
    else if (ta && !tb) GGAN_GEMM_CASE(true, false, 0, "<true, false>");
    else GGAN_GEMM_CASE(true, true, 0, "<true, true>");
#undef GGAN_GEMM_CASE
    return 0;
}

static int gemm_launch(int ta, int tb, int M, int N, int K, const float* A, const float* B, const float* bias, float* C,
                       float* colsum, int act, float alpha, void* ws, size_t ws_bytes, hipStream_t s,
                       const float* a_ref = nullptr, const float* b_ref = nullptr, int ref_act = 0, float ref_alpha = 0.f,
                       const float* A2 = nullptr, int a_split = 0, float* C2 = nullptr, int c_split = 0) {
    if (N == 1 && !ta && !tb && !colsum && !a_ref && !b_ref) {
        GGAN_LAUNCH("gemv_rows_k", 2.0 * M * K, 0, gemv_rows_k, dim3(cdiv(M, 4)), dim3(256), 0, s, A, B, bias, C, M, K, act, alpha);
        return 0;
    }
    // skinny products (gemm_skinny_k): where the 64x64 tiling would have to split K to fill the chip and K is short enough for all
    // of a wave's loads to be in flight at once -- one launch instead of product + reduce
    static const int skinny = [] { const char* e = getenv("GGAN_GEMM_SKINNY"); return e ? atoi(e) : 1; }();
    // (M >= 32: the scripts' minibatches are 50-128 rows.  The 8-row trajectory fixture `vegan-wgan-gp` is bimodal under fp32 rounding --
    //  a pre-activation within rounding of its LeakyReLU kink, profiles/r04_notes.md -- and ANY other legal summation order of its
    //  products, this one or GGAN_GEMM_SK=16 on the old kernel, lands it on the other branch: products that small keep their order)
    if (skinny && !ta && !colsum && (!A2 || (a_split % 16 == 0 && !a_ref)) && (!C2 || c_split % 16 == 0) && !b_ref && (!a_ref || tb) && M >= 32 &&
        K >= 64 && K <= 1024 && cdiv(M, BM) * cdiv(N, BN) < 64 &&
        cdiv(M, 16) * cdiv(N, 16) <= 1024 && !(K <= 128 && cdiv(M, BM) * cdiv(N, BN) >= 32) && (size_t)M * K * 4 < 0x7FFFFFF0ull &&
        (size_t)N * K * 4 < 0x7FFFFFF0ull) {
        SkinnyParams P;
        memset(&P, 0, sizeof(P));
        P.A = A; P.B = B; P.bias = bias; P.C = C; P.a_ref = a_ref;
        P.M = M; P.N = N; P.K = K; P.lda = A2 ? a_split : K; P.ldb = tb ? K : N;
        P.act = act; P.alpha = alpha; P.ref_act = ref_act; P.ref_alpha = ref_alpha;
        P.A2 = A2; P.a_split = a_split; P.C2 = C2; P.c_split = c_split;
        P.vecA = al16(A) && (P.lda % 4 == 0) && (!a_ref || al16(a_ref));
        P.vecA2 = A2 && al16(A2) && ((K - a_split) % 4 == 0);
        P.vecB = tb && al16(B) && (K % 4 == 0);
        P.a_bytes = (unsigned)((size_t)M * P.lda * 4); P.a2_bytes = A2 ? (unsigned)((size_t)M * (K - a_split) * 4) : 0u;
        P.b_bytes = (unsigned)((size_t)N * K * 4);
        const int steps = cdiv(cdiv(K, 16), 8);
        const int ns = steps <= 1 ? 1 : (steps <= 2 ? 2 : (steps <= 4 ? 4 : 8));
        const dim3 grid(cdiv(N, 16), cdiv(M, 16));
        const double fl = 2.0 * M * N * (double)K;
#define GGAN_SKINNY_NS(TB_, MK_, NAME_)                                                                                              \
        do {                                                                                                                         \
            if (ns == 1) { GGAN_LAUNCH("gemm_skinny_k" NAME_, fl, 0, (gemm_skinny_k<TB_, MK_, 1>), grid, dim3(512), 0, s, P); }      \
            else if (ns == 2) { GGAN_LAUNCH("gemm_skinny_k" NAME_, fl, 0, (gemm_skinny_k<TB_, MK_, 2>), grid, dim3(512), 0, s, P); } \
            else if (ns == 4) { GGAN_LAUNCH("gemm_skinny_k" NAME_, fl, 0, (gemm_skinny_k<TB_, MK_, 4>), grid, dim3(512), 0, s, P); } \
            else { GGAN_LAUNCH("gemm_skinny_k" NAME_, fl, 0, (gemm_skinny_k<TB_, MK_, 8>), grid, dim3(512), 0, s, P); }              \
        } while (0)
        if (a_ref) GGAN_SKINNY_NS(true, true, "<true, true>");
        else if (tb) GGAN_SKINNY_NS(true, false, "<true, false>");
        else GGAN_SKINNY_NS(false, false, "<false, false>");
#undef GGAN_SKINNY_NS
        return 0;
    }
    GemmPlan G;
    int rc = gemm_plan(G, 0, ta, tb, M, N, K, A, B, bias, C, colsum, act, alpha, ws, ws_bytes, a_ref, b_ref, ref_act, ref_alpha, A2,
                       a_split, C2, c_split);
    if (rc) return rc;
    rc = gemm_launch_planned(G, ta, tb, s);
    if (rc) return rc;
    if (G.P.SK > 1)
        return launch_splitk_reduce((const float*)G.ws, G.P.SK, G.P.out_elems, C, bias, N, 1, act, alpha, s, G.P.slab_stride, colsum,
                                    colsum ? (size_t)N : 0);
    return 0;
}

// Linear + LeakyReLU + Linear(H -> 1): the tail of the joint critic (gan_inference_cifar10.py:244-255: zx1 on
// tf.concat([conv features, latent features]) -> lrelu -> Output) and of the code-space critics (Hyper3 -> HyperOutput,
// gmgan_inference_cifar10.py:282-301).  Forward = the split-K GEMM of the first Linear with its slabs left in the workspace +
// ONE tail kernel (slab sum, bias, LeakyReLU, the H -> 1 product) instead of reduce + gemv launches.
int ggan_critic_head_fwd(int M, int K1, int K2, int H, const float* a1, const float* a2, const float* w, const float* b,
                         const float* w_out, const float* b_out, float alpha, float* h, float* logits, void* ws, size_t ws_bytes,
                         ggan_stream_t stream) {
    GGAN_CHECK_ARG(a1 && w && b && w_out && b_out && h && logits, "null pointer");
    GGAN_CHECK_ARG(M > 0 && K1 > 0 && K2 >= 0 && H > 0 && (H % 4) == 0 && (a2 || K2 == 0), "bad shape");
    GGAN_CHECK_ARG(al16(b) && al16(w_out) && al16(h), "b, w_out, h must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    GemmPlan G;
    // (twice the workgroups of the other split products: the tail kernel below sums the slabs anyway, and this product sits alone on
    //  the step's critical chain -- 9 serial k-steps per workgroup at 256, 5 at 512: 1.127 -> 1.122 ms; 1024: slower again)
    int rc = gemm_plan(G, 0, 0, 0, M, H, K1 + K2, a1, w, nullptr, h, nullptr, GGAN_ACT_NONE, 0.f, ws, ws_bytes, nullptr, nullptr, 0, 0.f,
                       K2 ? a2 : nullptr, K2 ? K1 : 0, nullptr, 0, head_split_target());
    if (rc) return rc;
    rc = gemm_launch_planned(G, 0, 0, s);
    if (rc) return rc;
    const float* part = G.P.SK > 1 ? (const float*)G.ws : h;          // (SK == 1: the product already sits in h; summed "slab" of one)
    GGAN_LAUNCH("head_out_fwd_k", 2.0 * M * H, 4.0 * M * H * (G.P.SK + 1), head_out_fwd_k, dim3(cdiv(M, 2)), dim3(256), 0, s, part,
                G.P.SK, G.P.slab_stride, b, w_out, b_out, alpha, h, logits, M, H);
    return 0;
}

static int head_terms(HeadTerms& T, int M, int kind, int nterms, const int* rows, const float* labels, const float* weights,
                      const float* const* ext) {
    memset(&T, 0, sizeof(T));
    if (nterms < 1 || nterms > 4 || !rows || !labels || !weights || kind < 0 || kind > 1) return -1;
    int tot = 0;
    for (int k = 0; k < nterms; ++k) {
        if (rows[k] <= 0) return -1;
        T.rows[k] = rows[k]; T.z[k] = labels[k]; T.w[k] = weights[k];
        T.ext[k] = ext ? ext[k] : nullptr;
        if (!T.ext[k]) tot += rows[k];
    }
    T.count = nterms; T.kind = kind;
    return tot == M ? 0 : -1;
}

// ggan_critic_head_fwd for a head whose logits are known to feed ONE sigmoid-cross-entropy cost with these terms (consecutive row
// ranges of the logits: rows, label, weight; the cost = sum_k w_k * mean(bce(logits of term k, z_k)), gan_inference.py:104-117): the
// tail kernel also leaves g[M] = d cost / d logits for a unit upstream gradient and gh[M,H] = g w_out^T lrelu'(h), so that
// ggan_critic_head_bwd_tail can follow at once.  H <= 2048.
int ggan_critic_head_fwd_bce(int M, int K1, int K2, int H, const float* a1, const float* a2, const float* w, const float* b,
                             const float* w_out, const float* b_out, float alpha, float* h, float* logits, int kind, int nterms,
                             const int* term_rows, const float* labels, const float* weights, float* g, float* gh, void* ws,
                             size_t ws_bytes, ggan_stream_t stream) {
    GGAN_CHECK_ARG(a1 && w && b && w_out && b_out && h && logits && g && gh, "null pointer");
    GGAN_CHECK_ARG(M > 0 && K1 > 0 && K2 >= 0 && H > 0 && (H % 4) == 0 && H <= 2048 && (a2 || K2 == 0), "bad shape");
    GGAN_CHECK_ARG(al16(b) && al16(w_out) && al16(h) && al16(gh), "b, w_out, h, gh must be 16-byte aligned");
    HeadTerms T;
    GGAN_CHECK_ARG(head_terms(T, M, kind, nterms, term_rows, labels, weights, nullptr) == 0, "terms must be 1..4 row ranges covering the M rows");
    hipStream_t s = (hipStream_t)stream;
    GemmPlan G;
    int rc = gemm_plan(G, 0, 0, 0, M, H, K1 + K2, a1, w, nullptr, h, nullptr, GGAN_ACT_NONE, 0.f, ws, ws_bytes, nullptr, nullptr, 0, 0.f,
                       K2 ? a2 : nullptr, K2 ? K1 : 0, nullptr, 0, head_split_target());
    if (rc) return rc;
    rc = gemm_launch_planned(G, 0, 0, s);
    if (rc) return rc;
    const float* part = G.P.SK > 1 ? (const float*)G.ws : h;
    GGAN_LAUNCH("head_out_fwd_bce_k", 5.0 * M * H, 4.0 * M * H * (G.P.SK + 2), head_out_fwd_bce_k, dim3(cdiv(M, 2)), dim3(256), 0, s, part,
                G.P.SK, G.P.slab_stride, b, w_out, b_out, alpha, h, logits, M, H, T, g, gh);
    return 0;
}

// Backward of the same head from g = d cost / d logits [M]:
//   gh = g w_out^T * lrelu'(h)                      (scratch [M,H], caller-owned: also what the two products below consume)
//   d_wout[H] = h^T g, d_bout = sum g               (NULL: not wanted)
//   d_w[K1+K2,H] = [a1|a2]^T gh, d_b[H] = colsum gh (d_w NULL in generator steps: the critic's weights are not in the var_list)
//   [d_a1 | d_a2] = gh w^T                          (NULL: the inputs need no gradient)
// head kernel + ONE grouped launch for the two products (separate launches when an operand does not take the FAST loads).
static int critic_head_bwd_impl(int M, int K1, int K2, int H, const float* g, const float* a1, const float* a2, const float* w, const float* h,
                                const float* w_out, float alpha, float* gh, float* d_a1, float* d_a2, float* d_w, float* d_b, float* d_wout,
                                float* d_bout, void* ws, size_t ws_bytes, ggan_stream_t stream, const HeadTail* tail) {
    GGAN_CHECK_ARG(a1 && w && h && w_out && gh, "null pointer");
    GGAN_CHECK_ARG(M > 0 && K1 > 0 && K2 >= 0 && H > 0 && (a2 || K2 == 0), "bad shape");
    GGAN_CHECK_ARG(!d_a1 || K2 == 0 || d_a2, "d_a2 missing");
    GGAN_CHECK_ARG(!d_b || d_w, "d_b comes out of the weight-gradient product");
    hipStream_t s = (hipStream_t)stream;
    const int K = K1 + K2;
    if (g) {        // (NULL: gh was produced together with the cost, ggan_bce_head_bwd)
        GGAN_LAUNCH("head_out_bwd_k", 3.0 * M * H, 8.0 * M * H, head_out_bwd_k, dim3(cdiv(H, 16)), dim3(256), 0, s, g, h, w_out, alpha, gh,
                    d_wout, d_bout, M, H);
    }
    GemmPlan Gw, Ga;
    int nw = 0, na = 0;
    if (d_w) {      // d_w[K,H] = [a1|a2]^T gh: A stored [M,K] read transposed, sources split the OUTPUT rows; column sums of gh -> d_b
        int rc = gemm_plan(Gw, 1, 1, 0, K, H, M, a1, gh, nullptr, d_w, d_b, GGAN_ACT_NONE, 0.f, ws, ws_bytes, nullptr, nullptr, 0, 0.f,
                           K2 ? a2 : nullptr, K2 ? K1 : 0);
        if (rc) return rc;
        nw = 1;
    }
    if (d_a1) {     // [d_a1 | d_a2] = gh w^T: B = w stored [K,H] read transposed, output columns split at K1
        int rc = gemm_plan(Ga, 1, 0, 1, M, K, H, gh, w, nullptr, d_a1, nullptr, GGAN_ACT_NONE, 0.f, ws, ws_bytes, nullptr, nullptr, 0, 0.f,
                           nullptr, 0, K2 ? d_a2 : nullptr, K2 ? K1 : 0);
        if (rc) return rc;
        na = 1;
    }
    if (nw && na && Gw.P.fast == 1 && Ga.P.fast == 1 && !getenv("GGAN_NO_GEMM_GROUP")) {
        GemmGroup GG;
        memset(&GG, 0, sizeof(GG));
        // (the data-gradient product first: its workgroups walk K = H in 16 serial steps, the weight-gradient's only M / 32 --
        //  dispatched first, the long pole starts at once and the short workgroups fill in around it)
        GG.n = 2;
        GG.kind[0] = 1; GG.gx[0] = Ga.gx; GG.p[0] = Ga.P; GG.first[0] = 0;
        GG.kind[1] = 0; GG.gx[1] = Gw.gx; GG.p[1] = Gw.P; GG.first[1] = Ga.gx * Ga.gy;
        GG.first[2] = GG.first[1] + Gw.gx * Gw.gy;
        int nwg = GG.first[2];
        if (tail) {
            GG.has_tail = 1;
            GG.tail = *tail;
            GG.tail.first = nwg;
            nwg += 1 + cdiv(H, 16);
        }
        GGAN_LAUNCH_GROUP(4.0 * M * K * (double)H, nwg, s, GG);
        return 0;
    }
    if (tail && na && !nw && Ga.P.fast == 1 && !getenv("GGAN_NO_GEMM_GROUP")) {
        // generator steps (the critic's weights are frozen: the data-gradient product alone): the cost still rides in its launch
        GemmGroup GG;
        memset(&GG, 0, sizeof(GG));
        GG.n = 1;
        GG.kind[0] = 1; GG.gx[0] = Ga.gx; GG.p[0] = Ga.P; GG.first[0] = 0;
        GG.first[1] = Ga.gx * Ga.gy;
        GG.has_tail = 1;
        GG.tail = *tail;
        GG.tail.first = GG.first[1];
        GGAN_LAUNCH_GROUP(2.0 * M * K * (double)H, GG.first[1] + 1 + cdiv(H, 16), s, GG);
        return 0;
    }
    if (tail) {
        HeadTail T = *tail;
        T.first = 0;
        GGAN_LAUNCH("head_tail_k", 3.0 * M * H, 4.0 * M * H, head_tail_k, dim3(1 + cdiv(H, 16)), dim3(256), 0, s, T);
    }
    if (nw) { int rc = gemm_launch_planned(Gw, 1, 0, s); if (rc) return rc; }
    if (na) { int rc = gemm_launch_planned(Ga, 0, 1, s); if (rc) return rc; }
    return 0;
}

int ggan_critic_head_bwd(int M, int K1, int K2, int H, const float* g, const float* a1, const float* a2, const float* w, const float* h,
                         const float* w_out, float alpha, float* gh, float* d_a1, float* d_a2, float* d_w, float* d_b, float* d_wout,
                         float* d_bout, void* ws, size_t ws_bytes, ggan_stream_t stream) {
    return critic_head_bwd_impl(M, K1, K2, H, g, a1, a2, w, h, w_out, alpha, gh, d_a1, d_a2, d_w, d_b, d_wout, d_bout, ws, ws_bytes, stream,
                                nullptr);
}

// The backward of a head run through ggan_critic_head_fwd_bce (g and gh are given): the two products, and -- as extra workgroups of
// their launch -- what the cost needs from all rows: loss[0] (the cost, bit-identical to ggan_bce_logits_multi_fwd on the same
// terms), d_wout[H] = h^T g, d_bout = sum g (NULL: not wanted).
int ggan_critic_head_bwd_tail(int M, int K1, int K2, int H, const float* a1, const float* a2, const float* w, const float* h,
                              const float* w_out, float alpha, const float* gh, float* d_a1, float* d_a2, float* d_w, float* d_b,
                              float* d_wout, float* d_bout, const float* logits, const float* g, int kind, int nterms, const int* term_rows,
                              const float* labels, const float* weights, const float* const* ext, float* loss, void* ws, size_t ws_bytes,
                              ggan_stream_t stream) {
    GGAN_CHECK_ARG(logits && g && gh, "null pointer");
    HeadTail T;
    memset(&T, 0, sizeof(T));
    GGAN_CHECK_ARG(head_terms(T.terms, M, kind, nterms, term_rows, labels, weights, ext) == 0,
                   "terms must be 1..4: row ranges covering the M rows, plus terms read elsewhere (ext)");
    T.logits = logits; T.g = g; T.h = h; T.d_wout = d_wout; T.d_bout = d_bout; T.loss = loss; T.M = M; T.H = H;
    return critic_head_bwd_impl(M, K1, K2, H, nullptr, a1, a2, w, h, w_out, alpha, const_cast<float*>(gh), d_a1, d_a2, d_w, d_b, d_wout,
                                d_bout, ws, ws_bytes, stream, &T);
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
