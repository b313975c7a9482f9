// What v_permlane16_swap_b32 does to a wavefront (gfx950): prints, for every lane,
// the two results of __builtin_amdgcn_permlane16_swap(x, x) with x = lane id.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* o) {
  const unsigned x = threadIdx.x;
  auto a = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  o[threadIdx.x] = a[0];
  o[64 + threadIdx.x] = a[1];
}
int main() {
  int* d;
  hipMalloc(&d, 128 * sizeof(int));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[128];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int r = 0; r < 2; r++) {
    printf("result %d:", r);
    for (int i = 0; i < 64; i++) printf(" %d", h[64 * r + i]);
    printf("\n");
  }
  return 0;
}
