// Probe: does workgroup b of a 1-D grid run on XCC (b mod 8)?  Prints the histogram of (blockIdx.x & 7, XCC_ID) pairs for a
// grid launched on an idle device and for one launched while another kernel is resident.  Build: hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_map(int *hist, int spin) {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(20, 0, 4)" : "=s"(xcc));
  if (threadIdx.x == 0) atomicAdd(&hist[(blockIdx.x & 7) * 16 + (xcc & 15)], 1);
  for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(64);
}
int main() {
  int *d; hipMalloc(&d, 128 * sizeof(int));
  for (int trial = 0; trial < 3; trial++) {
    hipMemset(d, 0, 128 * sizeof(int));
    int grid = trial == 0 ? 8 * 200 : (trial == 1 ? 8 * 200 + 3 : 5000);
    hipLaunchKernelGGL(k_map, dim3(grid), dim3(256), trial == 2 ? 65536 : 0, 0, d, trial == 2 ? 20 : 0);
    hipDeviceSynchronize();
    int h[128]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("grid %d (LDS %d): rows = blockIdx&7, cols = XCC_ID\n", grid, trial == 2 ? 65536 : 0);
    for (int r = 0; r < 8; r++) { for (int c = 0; c < 8; c++) printf("%6d", h[r * 16 + c]); printf("\n"); }
  }
  return 0;
}
