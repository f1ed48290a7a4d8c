// vlr_kernels_wide.hip — the wide build of the call kernel and of the AFD log filter (see kLdsSamples in vlr_plan.h): the same source
// with the per-sample arrays of the workgroup's LDS state sized for sixteen samples instead of eight, for scenarios with nine to
// sixteen samples (the reference has no limit on the number of samples, grammar/mod.rs:129-190).  The standard build keeps eight:
// its static LDS decides how many workgroups of the tumor-normal workloads fit a CU.  Everything lives in namespace vlr_wide; the
// exported symbols are vlr_launch_call_kernel_wide and vlr_launch_afd_kernel_wide.  (Pileups above the LDS budget stay flagged
// VLR_LOCUS_TOO_DEEP for such plans: there is no wide deep build.)
#define VLR_WIDE_BUILD 1
#define VLR_LDS_SAMPLES 16
#define vlr vlr_wide
#include "vlr_kernels.hip"
