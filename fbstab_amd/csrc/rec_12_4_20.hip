// One instance of the record kernel family (fb_record_kernel.h): its six kernels and
// the factory fbstab_hip.hip's instance table calls.
#include "fb_record_kernel.h"

FB_RECORD_INSTANCE(12, 4, 20, 1, "fbstab_mpc_r16_kernel<12,4,20>")
