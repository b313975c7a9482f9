// One instance of the record kernel family (fb_record_kernel.h): its six kernels and
// the factory fbstab_hip.hip's instance table calls.
#include "fb_record_kernel.h"

FB_RECORD_INSTANCE(24, 8, 32, 2, "fbstab_mpc_r32_kernel<24,8,32>")
