// Micro-benchmark for a multi-wave SSOR sweep: W compute waves of one workgroup take the dependent steps in turn.  A
// step is split into what does not depend on the previous W - 1 steps ("prep": the step's records, the gathers and the
// partial sum of the HEAD of every row, 8 G entries) and what does ("crit": L late gathers, L multiply-adds that
// continue the partial sum, the new y, one LDS store).  All waves meet at s_barrier once per phase; in every phase
// exactly one wave runs its crit, the others a part of their prep.  Shader cycles per phase via s_memtime.
// hipcc -O3 --offload-arch=gfx950 phase_probe.hip -o phase_probe && ./phase_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
template <class T> __device__ __forceinline__ T lds_ld(uint32_t a) { return *reinterpret_cast<const __attribute__((address_space(3))) T *>(a); }
template <class T> __device__ __forceinline__ void lds_st(uint32_t a, T v) { *reinterpret_cast<__attribute__((address_space(3))) T *>(a) = v; }

template <int G, int L>
struct Step {
  uint32_t head[8 * G > 0 ? 8 * G : 1], tail[L], mine;
  double hv[8 * G > 0 ? 8 * G : 1], tv[L], yh[8 * G > 0 ? 8 * G : 1];
  double r, invd, yold, acc;
};

// HELPERS extra waves run a copy loop global -> LDS between barriers (the record ring's producers)
template <int W, int G, int L, int HELPERS>
__global__ __launch_bounds__(64 * (W + HELPERS)) void phase_kernel(unsigned long long *out, const u32x4 *src, int iters, int active) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 16384; i += blockDim.x) reinterpret_cast<double *>(lds)[i] = 1.0 + 1e-9 * i;
  __syncthreads();
  constexpr int kStride = 96 * (G + 1) + 48;
  const uint32_t rec = 65536 + (uint32_t)(lane % active) * kStride;
  unsigned long long t0 = 0, t1 = 0;
  double sink = 0.0;
  if (wid >= W) {  // helper: one 4 KB chunk per phase
    int q = wid;
    for (int ph = 0; ph < iters * W + W - 1; ++ph) {
      const u32x4 *s = src + (size_t)(q & 1023) * 256 + lane;
      const u32x4 b0 = s[0], b1 = s[64], b2 = s[128], b3 = s[192];
      const uint32_t d = 100000 + (uint32_t)(q & 7) * 4096 + lane * 16;
      lds_st<u32x4>(d, b0); lds_st<u32x4>(d + 1024, b1); lds_st<u32x4>(d + 2048, b2); lds_st<u32x4>(d + 3072, b3);
      q += HELPERS;
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_s_barrier();
    }
    return;
  }
  Step<G, L> S;
  S.acc = 0.0; S.yold = 0.0; S.r = 1.0; S.invd = 0.5; S.mine = (uint32_t)(lane * 8 + wid * 512);
#pragma unroll
  for (int u = 0; u < L; ++u) { S.tail[u] = ((lane * 37 + u * 101) * 8) & 0x7ff8; S.tv[u] = 1e-3; }
#pragma unroll
  for (int u = 0; u < 8 * G; ++u) { S.head[u] = ((lane * 53 + u * 211) * 8) & 0x7ff8; S.hv[u] = 1e-3; S.yh[u] = 0.0; }
  // records: addresses of head + tail, values of head + tail, r / invd
  auto prep_records = [&](int it) {
    const u32x4 hdr = lds_ld<u32x4>(65536 - 16);
    const int nrows = __builtin_amdgcn_readfirstlane((int)hdr.x);
    const uint32_t rc = rec + (uint32_t)((it & 1) * 16) + (uint32_t)(nrows & 0);
#pragma unroll
    for (int j = 0; j < 2 * G; ++j) {
      const u32x4 c = lds_ld<u32x4>(rc + 16 * j);
      S.head[4 * j] = c.x & 0x7ff8; S.head[4 * j + 1] = c.y & 0x7ff8; S.head[4 * j + 2] = c.z & 0x7ff8; S.head[4 * j + 3] = c.w & 0x7ff8;
    }
#pragma unroll
    for (int j = 0; j < (L + 3) / 4; ++j) {
      const u32x4 c = lds_ld<u32x4>(rc + 32 * G + 16 * j);
      S.tail[4 * j] = c.x & 0x7ff8;
      if (4 * j + 1 < L) S.tail[4 * j + 1] = c.y & 0x7ff8;
      if (4 * j + 2 < L) S.tail[4 * j + 2] = c.z & 0x7ff8;
      if (4 * j + 3 < L) S.tail[4 * j + 3] = c.w & 0x7ff8;
    }
#pragma unroll
    for (int j = 0; j < 4 * G; ++j) {
      const f64x2 a2 = lds_ld<f64x2>(rc + 48 * G + 64 + 16 * j);
      S.hv[2 * j] = a2.x; S.hv[2 * j + 1] = a2.y;
    }
#pragma unroll
    for (int j = 0; j < (L + 1) / 2; ++j) {
      const f64x2 a2 = lds_ld<f64x2>(rc + 112 * G + 64 + 16 * j);
      S.tv[2 * j] = a2.x;
      if (2 * j + 1 < L) S.tv[2 * j + 1] = a2.y;
    }
    const f64x2 ri = lds_ld<f64x2>(rc + 8);
    S.r = ri.x; S.invd = ri.y;
    S.tail[0] = (uint32_t)(lane * 8 + ((wid + W - 1) % W) * 512);  // the previous step's result: a real dependence
  };
  auto prep_gather = [&]() {
#pragma unroll
    for (int u = 0; u < 8 * G; ++u) S.yh[u] = lds_ld<double>(S.head[u]);
    S.yold = lds_ld<double>(S.mine);
  };
  auto prep_chain = [&]() {
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < 8 * G; ++u) acc += S.hv[u] * S.yh[u];
    S.acc = acc;
  };
  auto crit = [&]() {
    double yt[L];
#pragma unroll
    for (int u = 0; u < L; ++u) yt[u] = lds_ld<double>(S.tail[u]);
    double acc = S.acc;
#pragma unroll
    for (int u = 0; u < L; ++u) acc += S.tv[u] * yt[u];
    if (lane < active) lds_st<double>(S.mine, S.yold + (0.5 * (S.r - acc)) * S.invd);
    sink += acc;
  };
  auto bar = [&]() {
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): my LDS store is done before the others are released
    __builtin_amdgcn_s_barrier();
  };
  // wave w: crit in phase w of every round, prep parts in the W - 1 phases after it
  for (int ph = 0; ph < wid; ++ph) bar();
  if (wid == 0) t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    crit();
    bar();
    if constexpr (W == 1) { prep_records(it); prep_gather(); prep_chain(); }
    if constexpr (W == 2) { prep_records(it); prep_gather(); prep_chain(); bar(); }
    if constexpr (W == 3) { prep_records(it); bar(); prep_gather(); prep_chain(); bar(); }
    if constexpr (W == 4) { prep_records(it); bar(); prep_gather(); bar(); prep_chain(); bar(); }
  }
  if (wid == 0) t1 = __builtin_amdgcn_s_memtime();
  for (int ph = wid; ph < W - 1; ++ph) bar();  // everybody has executed iters * W + (W - 1) barriers
  if (wid == 0 && lane == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)sink; }
}

template <int W, int G, int L, int HELPERS>
void run(const u32x4 *src, unsigned long long *out) {
  const int iters = 3000;
  for (int active : {64, 21}) {
    hipFuncSetAttribute((const void *)phase_kernel<W, G, L, HELPERS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    phase_kernel<W, G, L, HELPERS><<<1, 64 * (W + HELPERS), 140 * 1024>>>(out, src, 10, active);
    phase_kernel<W, G, L, HELPERS><<<1, 64 * (W + HELPERS), 140 * 1024>>>(out, src, iters, active);
    unsigned long long h[2];
    hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    std::printf("W %d compute waves, head %2d entries, tail %2d, %d helpers, rows %2d: %7.1f cycles per step\n", W, 8 * G, L, HELPERS, active,
                (double)h[0] / ((double)iters * (W == 1 ? 1 : W)));
  }
}

int main() {
  u32x4 *src; unsigned long long *out;
  hipMalloc(&src, 16 << 20); hipMemset(src, 0, 16 << 20);
  hipMalloc(&out, 64);
  run<1, 1, 6, 0>(src, out);
  run<2, 1, 6, 0>(src, out);
  run<3, 1, 6, 0>(src, out);
  run<4, 1, 6, 0>(src, out);
  run<3, 1, 6, 3>(src, out);
  run<3, 1, 8, 0>(src, out);
  run<1, 0, 14, 0>(src, out);
  run<2, 0, 14, 0>(src, out);
  run<3, 0, 14, 0>(src, out);
  run<3, 0, 14, 3>(src, out);
  run<3, 0, 16, 0>(src, out);
  run<4, 0, 16, 0>(src, out);
  run<3, 2, 8, 0>(src, out);
  run<1, 0, 16, 0>(src, out);
  run<2, 0, 16, 0>(src, out);
  run<1, 0, 24, 0>(src, out);
  run<2, 0, 24, 0>(src, out);
  run<3, 0, 24, 0>(src, out);
  run<2, 1, 16, 0>(src, out);
  return 0;
}
