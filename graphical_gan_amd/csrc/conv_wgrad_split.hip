// The role-split filter-gradient kernels (wgrad4_kernel<GW, true>, conv_wgrad.hip) in a translation unit of their own: compiled with
// -mllvm -amdgpu-mfma-vgpr-form (build.py EXTRA_FLAGS) so that a 256-register wave keeps its 200 accumulators in architectural VGPRs.
#define GGAN_WGRAD_SPLIT_TU 1
#include "conv_wgrad.hip"
