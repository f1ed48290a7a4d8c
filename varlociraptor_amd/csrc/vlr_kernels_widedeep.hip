// vlr_kernels_widedeep.hip — the deep launch (vlr_kernels_deep.hip) of the wide build (vlr_kernels_wide.hip): coefficient triples
// in the plan-owned HBM pool AND the wide build's limits, for pileups above the LDS budget in plans that run the wide kernels (nine to
// sixteen samples, more than four l2fc terms or nested ranges on a path).  Namespace vlr_widedeep; exports vlr_launch_call_kernel_widedeep.
#define VLR_DEEP_BUILD 1
#define VLR_WIDE_BUILD 1
#define VLR_LDS_SAMPLES 16
#define vlr vlr_widedeep
#include "vlr_kernels.hip"
