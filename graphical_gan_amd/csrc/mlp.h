// Internal: launchers of the fused 512-wide MLP chain (mlp_chain.hip), called by the C entry points in gemm.hip.
#pragma once
#include "common.h"

namespace ggan {

constexpr int kMlpHidden = 512;       // hidden width the chain kernels are built for
constexpr int kMlpMaxIn = 256;        // K1 + K2 of the first layer (16 output tiles of the last data-gradient product)

// floats of the transposed-weight scratch the forward launch fills for the backward launch
size_t mlp_chain_wt_floats(int K1, int K2);

// h1 = lrelu([x1 | x2] w1 + b1), h2 = lrelu(h1 w2 + b2), h3 = lrelu(h2 w3 + b3), logits = h3 w_out + b_out; all row-local: one launch,
// a workgroup per 8 rows.  wt (NULL: no backward will follow): mlp_chain_wt_floats() floats, receives w1^T | w2^T | w3^T.
// Returns 0, or < 0 with the error set.
int mlp_chain_fwd_launch(int M, int K1, int K2, const float* x1, const float* x2, const float* const w[3], const float* const b[3],
                         const float* w_out, const float* b_out, float alpha, float* const h[3], float* logits, float* wt, hipStream_t s);

// gh2 = (gh3 w3^T) * lrelu'(h2), gh1 = (gh2 w2^T) * lrelu'(h1), [dx1 | dx2] = gh1 w1^T from the masked gradient gh3 at the third layer's
// pre-activation.  gh2 / gh1 (NULL: not kept -- generator steps need no weight gradients) and dx1 / dx2 (NULL: inputs are data).
// wt: the scratch the forward launch of the same weights filled.
int mlp_chain_bwd_launch(int M, int K1, int K2, const float* gh3, const float* wt, const float* h1, const float* h2, float alpha,
                         float* gh2, float* gh1, float* dx1, float* dx2, hipStream_t s);

}  // namespace ggan
