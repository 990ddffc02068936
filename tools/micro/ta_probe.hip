// Micro-benchmark: cycles per 64-lane vector load instruction per CU, for L1-resident data.
// hipcc -O3 --offload-arch=gfx950 ta_probe.hip -o ta_probe && ./ta_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <typename T, int ACTIVE>
__global__ __launch_bounds__(256) void load_kernel(const T *src, T *out, int iters, int stride_elems) {
  const int lane = threadIdx.x & 63;
  // every wave walks a 16 KB window private to its workgroup: L1 hits after the first pass
  const T *p = src + (size_t)blockIdx.x * 2048 + lane;
  T acc{};
  if (lane < ACTIVE) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        T v = p[(i * 8 + u) * stride_elems & 1023];
        if constexpr (sizeof(T) == 4) acc += v;
        else if constexpr (sizeof(T) == 8) acc += v;
        else { acc.x += v.x; acc.y += v.y; }
      }
    }
  }
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

template <typename T, int ACTIVE>
void run(const char *name, int stride) {
  const int grid = 256 * 4, iters = 2000;
  T *src, *out;
  hipMalloc(&src, sizeof(T) * ((size_t)grid * 2048 + 4096));
  hipMemset(src, 0, sizeof(T) * ((size_t)grid * 2048 + 4096));
  hipMalloc(&out, sizeof(T) * (size_t)grid * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  load_kernel<T, ACTIVE><<<grid, 256>>>(src, out, 10, stride);
  hipEventRecord(e0);
  load_kernel<T, ACTIVE><<<grid, 256>>>(src, out, iters, stride);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double loads_per_cu = (double)grid * 4 * iters * 8 / 256.0;
  const double cyc = ms * 1e-3 * 2.4e9 / loads_per_cu;
  std::printf("%-28s stride %3d: %.3f ms, %.1f cycles (2.4 GHz) per wave-load per CU, %.1f B/clk/CU\n", name, stride, ms, cyc,
              (double)sizeof(T) * ACTIVE / cyc);
  hipFree(src); hipFree(out);
}

int main() {
  run<float, 64>("dword x64 lanes", 1);
  run<double, 64>("dwordx2 x64 lanes", 1);
  run<double2, 64>("dwordx4 x64 lanes", 1);
  run<double2, 32>("dwordx4 x32 lanes", 1);
  run<double, 2>("dwordx2 x2 lanes", 1);
  run<double, 64>("dwordx2 x64 lanes misaligned", 3);
  run<double2, 32>("dwordx4 x32 lanes misaligned", 3);
  return 0;
}
