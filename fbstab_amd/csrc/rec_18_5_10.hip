// One instance of the record kernel family (fb_record_kernel.h): its six kernels and
// the factory fbstab_hip.hip's instance table calls.
#include "fb_record_kernel.h"

FB_RECORD_INSTANCE(18, 5, 10, 2, "fbstab_mpc_r32_kernel<18,5,10>")
