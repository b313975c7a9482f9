// What v_permlane16_swap_b32 and v_permlane32_swap_b32 do to a wavefront (gfx950): prints, for every lane,
// the two results of __builtin_amdgcn_permlane16_swap(x, x) and of __builtin_amdgcn_permlane32_swap(x, x)
// with x = lane id (round 6: the second is what the dense kernel's wavefront sums use for the halves).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* o) {
  const unsigned x = threadIdx.x;
  auto a = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  o[threadIdx.x] = a[0];
  o[64 + threadIdx.x] = a[1];
  auto b = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  o[128 + threadIdx.x] = b[0];
  o[192 + threadIdx.x] = b[1];
}
int main() {
  int* d;
  hipMalloc(&d, 256 * sizeof(int));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[256];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int r = 0; r < 4; r++) {
    printf("%s result %d:", r < 2 ? "permlane16_swap" : "permlane32_swap", r & 1);
    for (int i = 0; i < 64; i++) printf(" %d", h[64 * r + i]);
    printf("\n");
  }
  return 0;
}
