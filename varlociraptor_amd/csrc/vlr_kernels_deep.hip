// vlr_kernels_deep.hip — the deep build of the call kernel (see VLR_DEEP in vlr_kernels.hip): the same source with the coefficient
// triples of a locus in a plan-owned HBM pool instead of LDS, for pileups above the LDS budget (VLR_LOCUS_TOO_DEEP of the
// LDS-resident kernel).  Everything lives in namespace vlr_deep; the only exported symbol is vlr_launch_call_kernel_deep.
#define VLR_DEEP_BUILD 1
#define vlr vlr_deep
#include "vlr_kernels.hip"
