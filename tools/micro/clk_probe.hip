// shader clock seen by a kernel that occupies one CU vs one that fills the GPU: s_memtime (shader cycles) against
// s_memrealtime (100 MHz) over a fixed spin of dependent VALU work.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(unsigned long long *out, int iters) {
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float x = threadIdx.x;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 64; ++u) x = x * 1.0001f + 0.5f;
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = (unsigned long long)x; }
}
int main() {
  unsigned long long *out, h[3];
  hipMalloc(&out, 64);
  for (int grid : {1, 1, 256, 2048, 1}) {
    for (int rep = 0; rep < 3; ++rep) {
      spin<<<grid, 256>>>(out, 20000);
      hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
      std::printf("grid %4d: %llu shader cycles in %llu ticks of 100 MHz -> %.3f GHz\n", grid, h[0], h[1], (double)h[0] / ((double)h[1] * 10.0));
    }
  }
  return 0;
}
