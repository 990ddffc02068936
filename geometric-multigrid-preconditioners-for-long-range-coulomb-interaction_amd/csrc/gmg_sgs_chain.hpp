// gmg_sgs_chain.hpp -- the SSOR wavefront sweep of gmg_sgs_phase.hpp with the four waves handing the dependent step to
// each other through ONE LDS word instead of meeting at s_barrier four times per step.
//
// Reference: LA::MPI::PreconditionSSOR with AdditionalData(0.5), /root/reference/src/step-50.cc:970-973 (same records, same
// arithmetic, same order as gmg_sgs_phase.hpp, gmg_sgs.hpp and oracle/gmg_oracle.c:smoother_apply_inverse: bit-identical).
//
// What the barrier version paid (profiles/r02_sgs_phase_cycles.txt): of ~2 900 cycles a wave spent per turn, 770 - 1 140 were
// drain + s_barrier + restart, four times per step, although only ONE dependence is real: step t needs the y of step t - 1.
// Here every wave runs its turn (records -> registers, copy of its next block, head sums and T2 products, dependent phase)
// at its own pace and the order of the dependent phases is kept by a counter `done` (index of the last finished step) in LDS:
//   * head / T2 of step t gather columns that step t - 1 does not write; steps <= t - 2 must be finished: the wave waits
//     for done >= t - 2 (normally long true);
//   * the dependent phase of step t polls `done` TOGETHER WITH its T1 gathers: {read done, gather T1} in one burst; the LDS
//     serves a wave's requests in order, so if `done` already shows t - 1 the gathers behind it saw the new y -- the
//     successful poll costs no extra round trip, the hand-over is one LDS write -> read (~130 cycles) instead of a barrier
//     (~320) plus the gathers (~100);
//   * then the chain as before (T1 multiply-adds, T2 adds, the new y), one LDS store of y, one LDS store of done = t.
// Every spin is bounded: a wave that waits longer than ~0.1 s raises the abort word, everybody leaves for the barrier at
// the end of the range, the host sees GMG_ERR_HIP (as in gmg_sgs.hpp).  The fifth wave still touches the record stream
// ahead of the copies (L2 hits); it paces itself by `done`.
//
// MEASURED (64 k atoms, cycle 4, level 1, MI355X, gpurun_out r3d / profiles/r03_sgs_chain_cycles.txt): bit-exact, never
// aborted -- and SLOWER than the barriers: 3.20 ms per sweep pair against 2.48 ms.  A turn spends 900 - 1 900 cycles in the
// two polls: the dependent chain is flag store -> next poll's read (a poll iteration with its gathers takes ~300 cycles,
// so ~150 on average until the store is seen) -> chain -> y store, ~700 cycles per step, where the hardware barrier releases
// all four waves within tens of cycles of the last arrival.  An LDS word is not a cheaper hand-over than s_barrier on this
// machine (round 2 had measured the same for two waves, profiles/r02_phase_probe.txt).  Kept as an experiment
// (-DGMG_EXPERIMENTS, option sgs_chain); the shipped library does not contain it.
#pragma once
#include "gmg_sgs_phase.hpp"

namespace gmg {

constexpr uint32_t kChSpinLimit = 1u << 21;  // polls of >= ~100 cycles each: > 0.1 s

namespace ch {

using ph::Rec;
using ph::Turn;
using ph::copy_to_lds;
using ph::ph_key;
using sw::lds_ld;
using sw::lds_st;
using sw::u32x4;
using sw::f64x2;

__device__ __forceinline__ int vol_ld32(uint32_t addr) {
  return *reinterpret_cast<const volatile __attribute__((address_space(3))) int *>(addr);
}
__device__ __forceinline__ double vol_ld64(uint32_t addr) {
  return *reinterpret_cast<const volatile __attribute__((address_space(3))) double *>(addr);
}
__device__ __forceinline__ void vol_st32(uint32_t addr, int v) {
  *reinterpret_cast<volatile __attribute__((address_space(3))) int *>(addr) = v;
}

// done >= want?  (one LDS word, wave-uniform).  false: the sweep was aborted.
__device__ __forceinline__ bool wait_done(uint32_t flag, int want) {
  for (uint32_t spins = 0;; ++spins) {
    const int d = __builtin_amdgcn_readfirstlane(vol_ld32(flag));
    if (__builtin_expect(d >= want, 1)) return true;
    __builtin_amdgcn_s_sleep(1);
    if (spins > kChSpinLimit) vol_st32(flag + 4, 1);
    if ((spins & 63u) == 63u && __builtin_amdgcn_readfirstlane(vol_ld32(flag + 4))) return false;
  }
}

// One step t of shape (G, L1, L2) by one wave, start to finish.  false: aborted.
template <int G, int L1, int L2, bool FWD, bool TIMED>
__device__ __forceinline__ bool turn(Turn &T, bool first, int t, const char *base, double *stream_d, uint32_t region, uint32_t flag, int lane, double omega) {
  constexpr int L = L1 + L2;
  constexpr uint32_t stride = (uint32_t)ph_stride(G, L);
  Rec<G, L> C;
  int l1s = L1;  // T1 slots this step really uses (multiple of 4): the dependent phase stops there
  unsigned long long m1 = 0;
#define CH_T(acc) if constexpr (TIMED) { m1 = __builtin_amdgcn_s_memtime(); T.acc += m1 - T.m0; T.m0 = m1; }
  // ---- records -> registers (the block was copied during the wave's previous turn; all but the newest vector-memory
  // operation -- the forward sweep's prefix store -- done = it has arrived)
  if (FWD && !first) __builtin_amdgcn_s_waitcnt(0x0f71);
  else __builtin_amdgcn_s_waitcnt(0x0f70);
  CH_T(c_wait)
  {
    C.nrows = (T.key >> 8) & 0xff;
    l1s = T.key >> 16;
    const u32x4 hdr = lds_ld<u32x4>(region);  // (about the wave's NEXT step: looked at after the reads below)
    const uint32_t rec = region + 16u + (uint32_t)min(lane, C.nrows - 1) * stride;
    const f64x2 ri = lds_ld<f64x2>(rec);
    const u32x4 q = lds_ld<u32x4>(rec + 16);
    C.r = ri.x; C.invd = ri.y;
    C.prefix = __hiloint2double((int)q.y, (int)q.x);
    C.my = q.z; C.aux = q.w;
#pragma unroll
    for (int j = 0; j < 4 * G; ++j) {
      const f64x2 a2 = lds_ld<f64x2>(rec + 32 + 16 * j);
      C.hv[2 * j] = a2.x; C.hv[2 * j + 1] = a2.y;
    }
#pragma unroll
    for (int j = 0; j < L / 2; ++j) {
      const f64x2 a2 = lds_ld<f64x2>(rec + 32 + 64 * G + 16 * j);
      C.tv[2 * j] = a2.x; C.tv[2 * j + 1] = a2.y;
    }
#pragma unroll
    for (int j = 0; j < 2 * G; ++j) {
      const u32x4 c = lds_ld<u32x4>(rec + 32 + 64 * G + 8 * L + 16 * j);
      C.ha[4 * j] = c.x; C.ha[4 * j + 1] = c.y; C.ha[4 * j + 2] = c.z; C.ha[4 * j + 3] = c.w;
    }
#pragma unroll
    for (int j = 0; j < L / 4; ++j) {
      const u32x4 c = lds_ld<u32x4>(rec + 32 + 96 * G + 8 * L + 16 * j);
      C.ta[4 * j] = c.x; C.ta[4 * j + 1] = c.y; C.ta[4 * j + 2] = c.z; C.ta[4 * j + 3] = c.w;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // every read of the region is done: it may be overwritten
    T.nx_bytes = (uint32_t)__builtin_amdgcn_readfirstlane((int)hdr.y); T.nx_off = (uint32_t)__builtin_amdgcn_readfirstlane((int)hdr.z);  // block t + 4
    T.key = __builtin_amdgcn_readfirstlane((int)hdr.w);
    CH_T(c_p1)
  }
  asm volatile("" ::: "memory");
  // ---- my next block (step t + 4) global -> LDS; it is read one turn on
  if (T.nx_bytes) copy_to_lds(base + T.nx_off, region, T.nx_bytes, lane);
  CH_T(c_copy)
  // ---- head and T2 read columns that step t - 1 does not write: everything up to step t - 2 must be in place
  if (!wait_done(flag, t - 2)) return false;
  CH_T(c_bar)
  {
    double yh[G > 0 ? 8 * G : 1];
#pragma unroll
    for (int k = 0; k < 8 * G; ++k) yh[k] = lds_ld<double>(C.ha[k]);
    C.yold = 0.0;
    if constexpr (!FWD) C.yold = lds_ld<double>(C.my);
    double y2[L2 > 0 ? L2 : 1];
#pragma unroll
    for (int k = 0; k < L2; ++k) y2[k] = lds_ld<double>(C.ta[L1 + k]);
    double acc = FWD ? 0.0 : C.prefix;
#pragma unroll
    for (int k = 0; k < 8 * G; ++k) acc += C.hv[k] * yh[k];
    asm volatile("" : "+v"(acc));  // formed here, not behind the poll (the compiler would sink the chain into the dependent phase)
    C.acc = acc;
#pragma unroll
    for (int k = 0; k < L2; ++k) {
      double pr = C.tv[L1 + k] * y2[k];
      asm volatile("" : "+v"(pr));
      C.tv[L1 + k] = pr;
    }
  }
  CH_T(c_p2)
  asm volatile("" ::: "memory");
  // ---- the dependent phase: poll `done` together with the T1 gathers (pieces of four, as many as the step uses)
  double yt[L1];
  for (uint32_t spins = 0;; ++spins) {
    const int dv = vol_ld32(flag);
#pragma unroll
    for (int q = 0; q < L1 / 4; ++q) {
      if (q > 0 && 4 * q >= l1s) break;
#pragma unroll
      for (int k = 4 * q; k < 4 * q + 4; ++k) yt[k] = vol_ld64(C.ta[k]);
    }
    if (__builtin_expect(__builtin_amdgcn_readfirstlane(dv) >= t - 1, 1)) break;  // (in-order LDS: the gathers behind this read saw step t - 1's y)
    if (spins > kChSpinLimit) vol_st32(flag + 4, 1);
    if ((spins & 63u) == 63u && __builtin_amdgcn_readfirstlane(vol_ld32(flag + 4))) return false;
  }
  CH_T(c_bar)
  {
    double acc = C.acc;
#pragma unroll
    for (int q = 0; q < L1 / 4; ++q) {
      if (q > 0 && 4 * q >= l1s) break;
#pragma unroll
      for (int k = 4 * q; k < 4 * q + 4; ++k) acc += C.tv[k] * yt[k];
    }
#pragma unroll
    for (int k = 0; k < L2; ++k) acc += C.tv[L1 + k];
    if (lane < C.nrows) {
      lds_st<double>(C.my, C.yold + (omega * (C.r - acc)) * C.invd);
      if constexpr (FWD) stream_d[C.aux] = acc;
    }
  }
  asm volatile("" ::: "memory");
  vol_st32(flag, t);  // behind the y store in this wave's LDS queue: whoever reads t finds the new y
  if constexpr (TIMED) { __builtin_amdgcn_s_waitcnt(0xc07f); }
  CH_T(c_crit)
#undef CH_T
  return true;
}

// The steps t = w, w + 4, ... of a range by compute wave w (the shape of a step travels in the header of the wave's
// previous block: one dispatch per change of shape).
template <bool FWD, bool TIMED>
__device__ __forceinline__ void sweep(const PhRange *R, const uint4 *tab, const char *stream, double *stream_d, uint32_t region, uint32_t flag, int w, int lane, double omega,
                                      unsigned long long *tp) {
  const int n = R->n_steps;
  const char *base = stream + R->stream_off;
  Turn T{};
  int t = w;
  if (t < n) {
    const uint4 e = tab[t];
    T.key = (int)e.z;
    copy_to_lds(base + e.x, region, e.y, lane);
  }
  if constexpr (TIMED) T.m0 = __builtin_amdgcn_s_memtime();
  bool ok = true;
  while (t < n && ok) {
#define CH_CASE(g, l1, l2) \
    case ph_key(g, l1, l2): \
      do { \
        ok = turn<g, l1, l2, FWD, TIMED>(T, t == w, t, base, stream_d, region, flag, lane, omega); \
        t += kPhWaves; \
      } while (ok && t < n && (T.key & 0xff) == ph_key(g, l1, l2));  /* (steps of one shape in a row: no dispatch in between) */ \
      break;
    switch (T.key & 0xff) {
      CH_CASE(0, 4, 0) CH_CASE(0, 4, 8) CH_CASE(0, 4, 16) CH_CASE(0, 4, 24) CH_CASE(0, 8, 0) CH_CASE(0, 8, 8)
      CH_CASE(0, 8, 16) CH_CASE(0, 8, 24) CH_CASE(0, 12, 0) CH_CASE(0, 12, 8) CH_CASE(0, 12, 16) CH_CASE(0, 12, 24)
      CH_CASE(0, 16, 0) CH_CASE(0, 16, 8) CH_CASE(0, 16, 16) CH_CASE(0, 20, 0) CH_CASE(0, 20, 8) CH_CASE(0, 20, 16)
      CH_CASE(0, 24, 0) CH_CASE(0, 24, 8) CH_CASE(0, 28, 0) CH_CASE(0, 28, 8) CH_CASE(1, 4, 0) CH_CASE(1, 4, 8)
      CH_CASE(1, 4, 16) CH_CASE(1, 4, 24) CH_CASE(1, 8, 0) CH_CASE(1, 8, 8) CH_CASE(1, 8, 16) CH_CASE(1, 12, 0)
      CH_CASE(1, 12, 8) CH_CASE(1, 12, 16) CH_CASE(1, 16, 0) CH_CASE(1, 16, 8) CH_CASE(1, 20, 0) CH_CASE(1, 20, 8)
      CH_CASE(1, 24, 0) CH_CASE(1, 28, 0) CH_CASE(2, 4, 0) CH_CASE(2, 4, 8) CH_CASE(2, 4, 16) CH_CASE(2, 8, 0)
      CH_CASE(2, 8, 8) CH_CASE(2, 12, 0) CH_CASE(2, 12, 8) CH_CASE(2, 16, 0) CH_CASE(2, 20, 0) CH_CASE(3, 4, 0)
      CH_CASE(3, 4, 8) CH_CASE(3, 8, 0) CH_CASE(3, 12, 0)
      default:  // (the host builds no other shape: the step cannot be taken -- give up loudly instead of stalling the others)
        vol_st32(flag + 4, 1);
        ok = false;
        break;
    }
#undef CH_CASE
  }
  if (TIMED && tp && w == 0 && lane == 0) { tp[0] = T.c_wait; tp[1] = T.c_p1; tp[2] = T.c_copy; tp[3] = T.c_p2; tp[4] = T.c_crit; tp[5] = T.c_bar; }
}

}  // namespace ch

__global__ __launch_bounds__(kPhThreads) void sgs_chain_kernel(SgsPhaseArgs a, int *abort_flag) {
  extern __shared__ __attribute__((aligned(16))) char lds[];  // at LDS address 0: [y slots][4 regions][junk 256][done, abort]
  double *ylds = reinterpret_cast<double *>(lds);
  const uint32_t ring0 = (uint32_t)a.y_slots * 8u;
  const uint32_t junk = ring0 + (uint32_t)kPhWaves * (uint32_t)kPhRegion, flag = junk + 256u;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r_begin = a.block_rng[a.block0 + blockIdx.x], r_end = a.block_rng[a.block0 + blockIdx.x + 1];
  for (int rg = r_begin; rg < r_end; ++rg) {
    const PhRange *Rp = a.ranges + rg;
    struct { int n_steps, ws_off, n_own, n_ws, backward; uint32_t pf_lead, pf_step, stream_bytes; int64_t stream_off; } R;
    R.n_steps = Rp->n_steps; R.ws_off = Rp->ws_off; R.n_own = Rp->n_own; R.n_ws = Rp->n_ws; R.backward = Rp->backward;
    R.pf_lead = Rp->pf_lead; R.pf_step = Rp->pf_step; R.stream_bytes = Rp->stream_bytes; R.stream_off = Rp->stream_off;
    const int32_t *ws = a.ws_ci + R.ws_off;
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    if (a.prof) t0 = __builtin_amdgcn_s_memtime();
    for (int k0 = tid; k0 < R.n_ws; k0 += 16 * kPhThreads) {  // sixteen independent gathers in flight per thread
      int ci[16];
      double v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) ci[j] = k0 + j * kPhThreads < R.n_ws ? ws[k0 + j * kPhThreads] : -1;
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = ci[j] >= 0 ? a.ycur[ci[j]] : 0.0;
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (ci[j] >= 0) ylds[k0 + j * kPhThreads] = v[j];
    }
    if (tid == 0) { ch::vol_st32(flag, -1); ch::vol_st32(flag + 4, 0); }  // nothing finished yet, not aborted
    __syncthreads();
    if (a.prof) t1 = __builtin_amdgcn_s_memtime();
    if (wid < kPhWaves) {
      double *stream_d = reinterpret_cast<double *>(a.stream);
      const uint32_t region = ring0 + (uint32_t)wid * (uint32_t)kPhRegion;
      unsigned long long *tp = a.prof ? a.prof + 12 * (size_t)rg + 4 : nullptr;
      const uint4 *tab = a.blk_tab + Rp->blk_tab;
      if (tp) {  // (the instrumented variant is code of its own: the production sweep carries no timer branches)
        if (R.backward) ch::sweep<false, true>(Rp, tab, a.stream, stream_d, region, flag, wid, lane, a.omega, tp);
        else ch::sweep<true, true>(Rp, tab, a.stream, stream_d, region, flag, wid, lane, a.omega, tp);
      } else {
        if (R.backward) ch::sweep<false, false>(Rp, tab, a.stream, stream_d, region, flag, wid, lane, a.omega, nullptr);
        else ch::sweep<true, false>(Rp, tab, a.stream, stream_d, region, flag, wid, lane, a.omega, nullptr);
      }
    } else {
      // prefetch wave: one 4-byte copy per 128-byte line into the junk area, pf_step bytes per finished step, pf_lead ahead
      const char *base = a.stream + R.stream_off;
      uint32_t cur = 0;
      for (uint32_t spins = 0;;) {
        const int d = __builtin_amdgcn_readfirstlane(ch::vol_ld32(flag));
        const uint32_t target = (uint32_t)min((unsigned long long)R.stream_bytes, (unsigned long long)R.pf_lead + (unsigned long long)(d + 2) * R.pf_step);
        const bool progressed = cur < target;
        for (; cur < target; cur += 8192)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + min(cur + (uint32_t)lane * 128u, R.stream_bytes - 4u)),
                                           (__attribute__((address_space(3))) void *)(uintptr_t)junk, 4, 0, 0);
        if (d >= R.n_steps - 1 || cur >= R.stream_bytes) break;
        spins = progressed ? 0 : spins + 1;
        if (spins > kChSpinLimit || __builtin_amdgcn_readfirstlane(ch::vol_ld32(flag + 4))) break;  // (the compute waves raise the abort word themselves)
        __builtin_amdgcn_s_sleep(16);
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): prefix stores, leftover copies
    __syncthreads();
    if (ch::vol_ld32(flag + 4)) {  // a wave gave up waiting: the sweep is broken (never seen; guards the GPU against a hang)
      if (tid == 0) *abort_flag = 1;
      return;
    }
    if (a.prof) t2 = __builtin_amdgcn_s_memtime();
    for (int k0 = tid; k0 < R.n_own; k0 += 16 * kPhThreads) {
      int ci[16], row[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) ci[j] = k0 + j * kPhThreads < R.n_own ? ws[k0 + j * kPhThreads] : -1;
#pragma unroll
      for (int j = 0; j < 16; ++j) row[j] = (R.backward && ci[j] >= 0) ? a.ci_row[ci[j]] : -1;
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (ci[j] >= 0) {
          const double v = ylds[k0 + j * kPhThreads];
          a.ycur[ci[j]] = v;
          if (row[j] >= 0) a.y[row[j]] = v;
        }
    }
    __syncthreads();
    if (a.prof) {
      t3 = __builtin_amdgcn_s_memtime();
      if (tid == 0) {
        unsigned long long *o = a.prof + 12 * (size_t)rg;
        o[0] = t2 - t1; o[1] = 0; o[2] = t1 - t0; o[3] = t3 - t2;
      }
    }
  }
}

}  // namespace gmg
