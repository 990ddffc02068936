// Micro-benchmark: what one wave alone on a CU pays for LDS instructions (latency of a dependent read, cycles per
// instruction of independent reads of 8 / 16 bytes per lane, of DP multiply-add chains), with the other three waves of
// the workgroup idle, polling an LDS word, or copying global memory into LDS.  Shader cycles via s_memtime.
// hipcc -O3 --offload-arch=gfx950 lds_probe.hip -o lds_probe && ./lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <class T> __device__ __forceinline__ T lds_ld(uint32_t a) { return *reinterpret_cast<const __attribute__((address_space(3))) T *>(a); }
template <class T> __device__ __forceinline__ void lds_st(uint32_t a, T v) { *reinterpret_cast<__attribute__((address_space(3))) T *>(a) = v; }

// others: 0 idle (exit at once), 1 poll an LDS word with s_sleep, 2 copy global -> LDS continuously
template <int TEST>
__global__ __launch_bounds__(256) void probe(unsigned long long *out, const u32x4 *src, int others, int iters, int active) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 16384; i += 256) reinterpret_cast<uint32_t *>(lds)[i] = (uint32_t)(((i * 8 + 64) & 0x7ff8));  // pointer chain in bytes
  if (tid == 0) reinterpret_cast<uint32_t *>(lds)[16384] = 0;
  __syncthreads();
  const uint32_t flag = 65536;
  if (wid != 0) {
    if (others == 1) {
      while (lds_ld<uint32_t>(flag) == 0) __builtin_amdgcn_s_sleep(2);
    } else if (others == 2) {
      int q = wid;
      while (__hip_atomic_load(reinterpret_cast<__attribute__((address_space(3))) uint32_t *>(flag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
        const u32x4 *s = src + (size_t)(q & 1023) * 256 + lane;
        const u32x4 b0 = s[0], b1 = s[64], b2 = s[128], b3 = s[192];
        const uint32_t d = 70000 + (uint32_t)(q & 7) * 4096 + lane * 16;
        lds_st<u32x4>(d, b0); lds_st<u32x4>(d + 1024, b1); lds_st<u32x4>(d + 2048, b2); lds_st<u32x4>(d + 3072, b3);
        q += 3;
      }
    }
    return;
  }
  unsigned long long t0 = 0, t1 = 0;
  double sink = 0.0;
  uint32_t p = lane * 8;
  if (lane < active) {
    if constexpr (TEST == 0) {  // dependent ds_read_b32 chain
      t0 = __builtin_amdgcn_s_memtime();
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) p = lds_ld<uint32_t>(p);
      }
      t1 = __builtin_amdgcn_s_memtime();
      sink = p;
    } else if constexpr (TEST == 1) {  // 16 independent ds_read_b64 (random-ish addresses), then use
      uint32_t a[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) a[u] = ((lane * 37 + u * 101) * 8) & 0x7ff8;
      t0 = __builtin_amdgcn_s_memtime();
      for (int i = 0; i < iters; ++i) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = lds_ld<double>(a[u]);
#pragma unroll
        for (int u = 0; u < 16; ++u) sink += v[u];
      }
      t1 = __builtin_amdgcn_s_memtime();
    } else if constexpr (TEST == 2) {  // 8 independent ds_read_b128 at lane * 240 + imm
      const uint32_t base = lane * 240;
      t0 = __builtin_amdgcn_s_memtime();
      for (int i = 0; i < iters; ++i) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = lds_ld<u32x4>(base + 16 * u);
#pragma unroll
        for (int u = 0; u < 8; ++u) sink += (double)(v[u].x + v[u].w);
      }
      t1 = __builtin_amdgcn_s_memtime();
    } else if constexpr (TEST == 3) {  // dependent DP chain: 16 x (mul, add)
      double x = 1.0 + lane, acc = 0.0;
      t0 = __builtin_amdgcn_s_memtime();
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += x * (acc + (double)u);
      }
      t1 = __builtin_amdgcn_s_memtime();
      sink = acc;
    } else if constexpr (TEST >= 5) {
      // emulation of one SSOR sub-step of G = TEST - 4 groups: 8 G random y gathers + own y, 4 G + 1 sixteen-byte value reads,
      // 10 sixteen-byte reads of the next sub-step's columns, the chain of 8 G multiply-adds, one y write
      constexpr int G = TEST - 4;
      uint32_t a[8 * G];
#pragma unroll
      for (int u = 0; u < 8 * G; ++u) a[u] = ((lane * 37 + u * 101) * 8) & 0x7ff8;
      const uint32_t rec = 40000 + lane * (96 * G + 48), mine = lane * 8;
      double carry = 0.0;
      t0 = __builtin_amdgcn_s_memtime();
      for (int i = 0; i < iters; ++i) {
        double yv[8 * G], av[8 * G];
#pragma unroll
        for (int u = 0; u < 8 * G; ++u) yv[u] = lds_ld<double>(a[u]);
        const double yold = lds_ld<double>(mine);
#pragma unroll
        for (int j = 0; j < 4 * G; ++j) {
          typedef double f64x2 __attribute__((ext_vector_type(2)));
          const f64x2 a2 = lds_ld<f64x2>(rec + 32 + 16 * j);
          av[2 * j] = a2.x; av[2 * j + 1] = a2.y;
        }
        u32x4 nx[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) nx[j] = lds_ld<u32x4>(rec + 16 * j + (i & 1) * 16);
        __builtin_amdgcn_sched_barrier(0);
        double acc = carry;
#pragma unroll
        for (int u = 0; u < 8 * G; ++u) acc += av[u] * yv[u];
        carry = acc * 1e-30;
        lds_st<double>(mine, yold + acc * 1e-30);
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = (a[u] + (nx[u].x & 8)) & 0x7ff8;  // the prefetched columns feed the next gathers
        sink += (double)(nx[8].x + nx[9].y);
      }
      t1 = __builtin_amdgcn_s_memtime();
      sink += carry;
    } else if constexpr (TEST == 4) {  // write then dependent read of the same address (y hand-over between steps)
      t0 = __builtin_amdgcn_s_memtime();
      double v = lane;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { lds_st<double>(lane * 8, v); v = lds_ld<double>(lane * 8) + 1.0; }
      }
      t1 = __builtin_amdgcn_s_memtime();
      sink = v;
    }
  }
  lds_st<uint32_t>(flag, 1u);
  if (lane == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)sink; }
}

template <int TEST>
void run(const char *name, double ops_per_iter, const u32x4 *src, unsigned long long *out) {
  for (int active : {64, 21})
    for (int others = 0; others < 3; ++others) {
      const int iters = 2000;
      hipFuncSetAttribute((const void *)probe<TEST>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      probe<TEST><<<1, 256, 140 * 1024>>>(out, src, others, 10, active);
      probe<TEST><<<1, 256, 140 * 1024>>>(out, src, others, iters, active);
      unsigned long long h[2];
      hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
      std::printf("%-44s active lanes %2d, other waves %s: %7.1f cycles per op\n", name, active, others == 0 ? "idle   " : others == 1 ? "polling" : "copying",
                  (double)h[0] / (iters * ops_per_iter));
    }
}

int main() {
  u32x4 *src; unsigned long long *out;
  hipMalloc(&src, 16 << 20); hipMemset(src, 0, 16 << 20);
  hipMalloc(&out, 64);
  run<0>("dependent ds_read_b32 (latency)", 16, src, out);
  run<1>("16 independent ds_read_b64 + 16 adds", 16, src, out);
  run<2>("8 independent ds_read_b128 + use", 8, src, out);
  run<3>("dependent DP mul+add pair", 16, src, out);
  run<4>("ds_write_b64 -> ds_read_b64 same address", 16, src, out);
  run<6>("sub-step emulation, 2 groups (16 entries)", 1, src, out);
  run<7>("sub-step emulation, 3 groups (24 entries)", 1, src, out);
  run<8>("sub-step emulation, 4 groups (32 entries)", 1, src, out);
  return 0;
}
