// vlr_kernels_wide.hip — the wide build of the call kernel and of the AFD log filter (see kLdsSamples in vlr_plan.h): the same source
// with the per-sample arrays of the workgroup's LDS state sized for sixteen samples instead of eight and room for eight l2fc terms and
// eight nested ranges on a path instead of four (vlr_plan.h), for scenarios beyond the standard build's limits (the reference has
// none: grammar/mod.rs:129-190, grammar/vaftree.rs:168-305).  The standard build keeps the small arrays: its static LDS decides how
// many workgroups of the tumor-normal workloads fit a CU.  Everything lives in namespace vlr_wide; the exported symbols are
// vlr_launch_call_kernel_wide, vlr_launch_afd_kernel_wide and vlr_plan_lds_floor_wide.  (Pileups above the LDS budget of such plans:
// vlr_kernels_widedeep.hip.)
#define VLR_WIDE_BUILD 1
#define VLR_LDS_SAMPLES 16
#define vlr vlr_wide
#include "vlr_kernels.hip"
