// gmg_sgs_reg.hpp -- the four-wave SSOR sweep of gmg_sgs_phase.hpp with the records loaded from global memory STRAIGHT INTO
// REGISTERS: no LDS staging regions, no copy phase, no phase that reads the records back.
//
// Reference: LA::MPI::PreconditionSSOR with AdditionalData(0.5), /root/reference/src/step-50.cc:970-973 (same arithmetic, same
// order as gmg_sgs_phase.hpp, gmg_sgs.hpp and oracle/gmg_oracle.c:smoother_apply_inverse: bit-identical results).
//
// Why.  The phase table of the four-wave sweep (profiles/r03_sgs_phase_cycles.txt) has, per step of ~620 cycles, the
// longest phase NOT in the dependent part: P1 = 270 cycles waiting for the wave's LDS copy + 260 reading the records back,
// COPY = 450 (a global_load_lds costs ~45 cycles to issue), against CRIT = 380 forward / 530 backward.  Both exist only
// because the records travel global -> LDS -> registers.  With the records laid out FIELD-major (unit k of row u at
// block + 16 + (k * nrows + u) * 16: the layout gmg_sgs_dep.hpp introduced) one ordinary 16-byte load per unit brings a
// field of all rows of the step into the registers that will use it -- issued two phases before they are needed, into the
// registers the wave's previous step has just released (one register set), ~20 cycles per instruction.  A wave's turn:
//     phase t       LOAD  the step's record, global -> registers (+ the block's header: where the wave's next block is)
//     phase t + 1   --    (the loads are in flight)
//     phase t + 2   P2    head: gathers + partial sum; T2: gathers + products           (as gmg_sgs_phase.hpp)
//     phase t + 3   CRIT  T1 gathers, T1 multiply-adds, T2 adds, the new y, one LDS store (as gmg_sgs_phase.hpp)
// The 64 KB of staging regions go to the y slots: 20 416 doubles instead of 12 256, fewer ranges, fewer working-set loads.
//
// Status: OPT-IN (option sgs_reg=1), bit-identical, and SLOWER: 3.5-4.6 ms against 2.48 ms per level-1 application at 64 k
// atoms (gpurun_out/r3h, round 3).  The instrumented variant has the LOAD phase at 950-1450 cycles per turn with no wait
// in its instruction stream (checked in the ISA: SGPR base + one VGPR offset per load, no s_waitcnt before the barrier):
// ISSUING ~20-30 one-KB loads takes that long.  One CU pulls a stream it misses in its L1 at ~20 bytes per clock (outstanding
// lines / latency), whatever the prefetch wave does (no prefetch: +10 %; a lead of 512 KB or touches every 64 bytes:
// no change) -- the same ~45 cycles per KB the LDS copies of gmg_sgs_phase.hpp pay, but there the copy has a phase of its
// own and two more to land, here it holds the wave.  What this says about the four-wave sweep: its record STREAM is the
// thing to shrink (gmg_sgs_phase.hpp: records stored at the step's own width).
#pragma once
#include "gmg_sgs_dep.hpp"

namespace gmg {

constexpr int kRgYSlots = (160 * 1024 - kPhJunk) / 8 & ~1;  // doubles of y in LDS: 20 416

namespace rg {

using ph::bar;
using ph::ph_key;
using sw::lds_ld;
using sw::lds_st;
using sw::u32x4;
using sw::f64x2;

// The record of a step, field-major, into registers.  Every address is (uniform base in SGPRs) + (the lane's 32-bit offset, ONE
// VGPR for the whole record): with per-lane 64-bit pointers the register allocator recycles a pointer's registers as the
// destination of a load and then has to wait for that load before it can form the next pointer (seen in the ISA: an
// s_waitcnt vmcnt in the middle of the LOAD phase, i.e. a full memory latency on the phase's path).
template <int G, int L>
__device__ __forceinline__ void load_record(dp::Raw<G, L> &R, const char *blk, int nrows, int lane) {
  const uint32_t lo = (uint32_t)min(lane, nrows - 1) * 16u;
  const uint64_t b0 = (uint64_t)(blk + 16);
  const uint64_t ub = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b0 >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b0);
  const uint32_t us = (uint32_t)nrows * 16u;  // bytes between two units
  int u = 0;
  auto at = [&](int k) { return reinterpret_cast<const char *>(ub + (uint64_t)(us * (uint32_t)k)) + lo; };
  R.ri = *reinterpret_cast<const f64x2 *>(at(u++));
  R.q = *reinterpret_cast<const u32x4 *>(at(u++));
#pragma unroll
  for (int j = 0; j < 4 * G; ++j) R.hv[j] = *reinterpret_cast<const f64x2 *>(at(u++));
#pragma unroll
  for (int j = 0; j < L / 2; ++j) R.tv[j] = *reinterpret_cast<const f64x2 *>(at(u++));
#pragma unroll
  for (int j = 0; j < 2 * G; ++j) R.ha[j] = *reinterpret_cast<const u32x4 *>(at(u++));
#pragma unroll
  for (int j = 0; j < L / 4; ++j) R.ta[j] = *reinterpret_cast<const u32x4 *>(at(u++));
}

struct Turn {
  uint32_t off;   // the wave's next block in the range's stream
  int key;        // its shape | rows << 8 | T1 slots in use << 16   (key < 0: none)
  unsigned long long c_wait, c_load, c_p2, c_crit, c_bar, m0;
};

// One step of shape (G, L1, L2) by one wave: four phases, four barriers.
template <int G, int L1, int L2, bool FWD, bool TIMED>
__device__ __forceinline__ void turn(Turn &T, bool more, const char *base, double *stream_d, int lane, double omega) {
  constexpr int L = L1 + L2;
  unsigned long long m1 = 0;
#define RG_T(acc) if constexpr (TIMED) { m1 = __builtin_amdgcn_s_memtime(); T.acc += m1 - T.m0; T.m0 = m1; }
  // ---- LOAD: the record, field by field; the header tells where the wave's next block is
  const int nrows = (T.key >> 8) & 0xff, l1s = T.key >> 16;
  const char *blk = base + T.off;
  // the block's header (where the wave's next block is, and its shape) comes with the record: one more vector load, every lane
  // the same 16 bytes; it is looked at when the turn ends (a scalar load here would sit in lgkmcnt: ~1400 cycles on a miss, and
  // every barrier below waits for lgkmcnt(0))
  // (only the two words that are used: a dead half of the destination would be recycled for addresses while the load is
  // in flight, which costs a wait for it)
  uint2 hdr = *reinterpret_cast<const uint2 *>(blk + 8);
  dp::Raw<G, L> R;
  load_record(R, blk, nrows, lane);
  RG_T(c_load)
  bar();
  RG_T(c_bar)
  bar();
  RG_T(c_bar)
  // ---- P2: the head (no column of it is written in this phase or the next) and the products behind the last late column
  if constexpr (TIMED) { __builtin_amdgcn_s_waitcnt(0x0f70); RG_T(c_wait) }
  double acc0, yold = 0.0;
  double pr[L2 > 0 ? L2 : 1];
  {
    double yh[G > 0 ? 8 * G : 1];
#pragma unroll
    for (int j = 0; j < 2 * G; ++j) {
      yh[4 * j] = lds_ld<double>(R.ha[j].x); yh[4 * j + 1] = lds_ld<double>(R.ha[j].y);
      yh[4 * j + 2] = lds_ld<double>(R.ha[j].z); yh[4 * j + 3] = lds_ld<double>(R.ha[j].w);
    }
    if constexpr (!FWD) yold = lds_ld<double>(R.q.z);
    double y2[L2 > 0 ? L2 : 1];
#pragma unroll
    for (int j = 0; j < L2 / 4; ++j) {
      const u32x4 a4 = R.ta[L1 / 4 + j];
      y2[4 * j] = lds_ld<double>(a4.x); y2[4 * j + 1] = lds_ld<double>(a4.y); y2[4 * j + 2] = lds_ld<double>(a4.z); y2[4 * j + 3] = lds_ld<double>(a4.w);
    }
    double acc = FWD ? 0.0 : __hiloint2double((int)R.q.y, (int)R.q.x);
#pragma unroll
    for (int j = 0; j < 4 * G; ++j) { acc += R.hv[j].x * yh[2 * j]; acc += R.hv[j].y * yh[2 * j + 1]; }
    asm volatile("" : "+v"(acc));  // formed here, not after the barrier (the compiler would sink the chain into CRIT)
    acc0 = acc;
#pragma unroll
    for (int j = 0; j < L2 / 2; ++j) {
      double p0 = R.tv[L1 / 2 + j].x * y2[2 * j], p1 = R.tv[L1 / 2 + j].y * y2[2 * j + 1];
      asm volatile("" : "+v"(p0), "+v"(p1));
      pr[2 * j] = p0; pr[2 * j + 1] = p1;
    }
  }
  RG_T(c_p2)
  bar();
  RG_T(c_bar)
  // ---- CRIT: from the first late column on (in pieces of four, left by one forward branch)
  {
    double yt[L1];
#pragma unroll
    for (int q = 0; q < L1 / 4; ++q) {
      if (q > 0 && 4 * q >= l1s) break;
      const u32x4 a4 = R.ta[q];
      yt[4 * q] = lds_ld<double>(a4.x); yt[4 * q + 1] = lds_ld<double>(a4.y); yt[4 * q + 2] = lds_ld<double>(a4.z); yt[4 * q + 3] = lds_ld<double>(a4.w);
    }
    double acc = acc0;
#pragma unroll
    for (int q = 0; q < L1 / 4; ++q) {
      if (q > 0 && 4 * q >= l1s) break;
      acc += R.tv[2 * q].x * yt[4 * q]; acc += R.tv[2 * q].y * yt[4 * q + 1];
      acc += R.tv[2 * q + 1].x * yt[4 * q + 2]; acc += R.tv[2 * q + 1].y * yt[4 * q + 3];
    }
#pragma unroll
    for (int k = 0; k < L2; ++k) acc += pr[k];
    if (lane < nrows) {
      lds_st<double>(R.q.z, yold + (omega * (R.ri.x - acc)) * R.ri.y);
      if constexpr (FWD) stream_d[R.q.w] = acc;
    }
  }
  if constexpr (TIMED) { __builtin_amdgcn_s_waitcnt(0xc07f); }
  RG_T(c_crit)
  bar();
  RG_T(c_bar)
  asm volatile("" : "+v"(hdr.x), "+v"(hdr.y));  // (not before: the compiler would wait for the loads in the LOAD phase)
  if (more) { T.off = (uint32_t)__builtin_amdgcn_readfirstlane((int)hdr.x); T.key = __builtin_amdgcn_readfirstlane((int)hdr.y); }
#undef RG_T
}

// The steps t = w, w + 4, ... of a range by compute wave w.
template <bool FWD, bool TIMED>
__device__ __forceinline__ void sweep(const PhRange *R, const uint4 *tab, const char *stream, double *stream_d, int w, int lane, double omega, unsigned long long *tp) {
  const int n = R->n_steps;
  const char *base = stream + R->stream_off;
  Turn T{};
  int t = w;
  if (t < n) {
    const uint4 e = tab[t];
    T.off = e.x; T.key = (int)e.z;
  }
  for (int i = 0; i < w; ++i) bar();
  int done = w;
  if constexpr (TIMED) T.m0 = __builtin_amdgcn_s_memtime();
  while (t < n) {
#define RG_CASE(g, l1, l2) \
    case ph_key(g, l1, l2): \
      do { \
        turn<g, l1, l2, FWD, TIMED>(T, t + kPhWaves < n, base, stream_d, lane, omega); \
        t += kPhWaves; done += kPhWaves; \
      } while (t < n && (T.key & 0xff) == ph_key(g, l1, l2));  /* (steps of one shape in a row: no dispatch in between) */ \
      break;
    switch (T.key & 0xff) {
      RG_CASE(0, 4, 0) RG_CASE(0, 4, 8) RG_CASE(0, 4, 16) RG_CASE(0, 4, 24) RG_CASE(0, 8, 0) RG_CASE(0, 8, 8)
      RG_CASE(0, 8, 16) RG_CASE(0, 8, 24) RG_CASE(0, 12, 0) RG_CASE(0, 12, 8) RG_CASE(0, 12, 16) RG_CASE(0, 12, 24)
      RG_CASE(0, 16, 0) RG_CASE(0, 16, 8) RG_CASE(0, 16, 16) RG_CASE(0, 20, 0) RG_CASE(0, 20, 8) RG_CASE(0, 20, 16)
      RG_CASE(0, 24, 0) RG_CASE(0, 24, 8) RG_CASE(0, 28, 0) RG_CASE(0, 28, 8) RG_CASE(1, 4, 0) RG_CASE(1, 4, 8)
      RG_CASE(1, 4, 16) RG_CASE(1, 4, 24) RG_CASE(1, 8, 0) RG_CASE(1, 8, 8) RG_CASE(1, 8, 16) RG_CASE(1, 12, 0)
      RG_CASE(1, 12, 8) RG_CASE(1, 12, 16) RG_CASE(1, 16, 0) RG_CASE(1, 16, 8) RG_CASE(1, 20, 0) RG_CASE(1, 20, 8)
      RG_CASE(1, 24, 0) RG_CASE(1, 28, 0) RG_CASE(2, 4, 0) RG_CASE(2, 4, 8) RG_CASE(2, 4, 16) RG_CASE(2, 8, 0)
      RG_CASE(2, 8, 8) RG_CASE(2, 12, 0) RG_CASE(2, 12, 8) RG_CASE(2, 16, 0) RG_CASE(2, 20, 0) RG_CASE(3, 4, 0)
      RG_CASE(3, 4, 8) RG_CASE(3, 8, 0) RG_CASE(3, 12, 0)
      default:  // (the host builds no other shape)
        for (int i = 0; i < kPhWaves; ++i) bar();
        t += kPhWaves; done += kPhWaves;
        break;
    }
#undef RG_CASE
  }
  for (; done < n + kPhWaves - 1; ++done) bar();
  if (TIMED && tp && w == 0 && lane == 0) { tp[0] = T.c_wait; tp[1] = T.c_load; tp[2] = 0; tp[3] = T.c_p2; tp[4] = T.c_crit; tp[5] = T.c_bar; }
}

}  // namespace rg

__global__ __launch_bounds__(kPhThreads) void sgs_regs_kernel(SgsPhaseArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];  // at LDS address 0: [y slots][junk]
  double *ylds = reinterpret_cast<double *>(lds);
  const uint32_t junk = (uint32_t)a.y_slots * 8u;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r_begin = a.block_rng[a.block0 + blockIdx.x], r_end = a.block_rng[a.block0 + blockIdx.x + 1];
  for (int rg_i = r_begin; rg_i < r_end; ++rg_i) {
    const PhRange *Rp = a.ranges + rg_i;
    const int n_steps = Rp->n_steps, n_own = Rp->n_own, n_ws = Rp->n_ws, backward = Rp->backward;
    const uint32_t pf_lead = Rp->pf_lead, pf_step = Rp->pf_step, stream_bytes = Rp->stream_bytes;
    const int32_t *ws = a.ws_ci + Rp->ws_off;
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    if (a.prof) t0 = __builtin_amdgcn_s_memtime();
    for (int k0 = tid; k0 < n_ws; k0 += 16 * kPhThreads) {  // sixteen independent gathers in flight per thread
      int ci[16];
      double v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) ci[j] = k0 + j * kPhThreads < n_ws ? ws[k0 + j * kPhThreads] : -1;
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = ci[j] >= 0 ? a.ycur[ci[j]] : 0.0;
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (ci[j] >= 0) ylds[k0 + j * kPhThreads] = v[j];
    }
    __syncthreads();
    if (a.prof) t1 = __builtin_amdgcn_s_memtime();
    if (wid < kPhWaves) {
      double *stream_d = reinterpret_cast<double *>(a.stream);
      unsigned long long *tp = a.prof ? a.prof + 12 * (size_t)rg_i + 4 : nullptr;
      const uint4 *tab = a.blk_tab + Rp->blk_tab;
      if (tp) {  // (the instrumented variant is code of its own: the production sweep carries no timer branches)
        if (backward) rg::sweep<false, true>(Rp, tab, a.stream, stream_d, wid, lane, a.omega, tp);
        else rg::sweep<true, true>(Rp, tab, a.stream, stream_d, wid, lane, a.omega, tp);
      } else {
        if (backward) rg::sweep<false, false>(Rp, tab, a.stream, stream_d, wid, lane, a.omega, nullptr);
        else rg::sweep<true, false>(Rp, tab, a.stream, stream_d, wid, lane, a.omega, nullptr);
      }
    } else {
      // prefetch wave: one 4-byte copy per 128-byte line, pf_step bytes per phase, into the junk area (the loads then hit the L2)
      const char *base = a.stream + Rp->stream_off;
      uint32_t cur = 0;
      const uint32_t first = min(pf_lead, stream_bytes);
      for (; cur < first; cur += 8192)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + min(cur + (uint32_t)lane * 128u, stream_bytes - 4u)),
                                         (__attribute__((address_space(3))) void *)(uintptr_t)junk, 4, 0, 0);
      for (int p = 0; p < n_steps + kPhWaves - 1; ++p) {
        const uint32_t end = min(cur + pf_step, stream_bytes);
        for (; cur < end; cur += 8192)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + min(cur + (uint32_t)lane * 128u, stream_bytes - 4u)),
                                           (__attribute__((address_space(3))) void *)(uintptr_t)junk, 4, 0, 0);
        __builtin_amdgcn_s_barrier();
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): prefix stores, leftover prefetches
    __syncthreads();
    if (a.prof) t2 = __builtin_amdgcn_s_memtime();
    for (int k0 = tid; k0 < n_own; k0 += 16 * kPhThreads) {
      int ci[16], row[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) ci[j] = k0 + j * kPhThreads < n_own ? ws[k0 + j * kPhThreads] : -1;
#pragma unroll
      for (int j = 0; j < 16; ++j) row[j] = (backward && ci[j] >= 0) ? a.ci_row[ci[j]] : -1;
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
        unsigned long long *o = a.prof + 12 * (size_t)rg_i;
        o[0] = t2 - t1; o[1] = 0; o[2] = t1 - t0; o[3] = t3 - t2;
      }
    }
  }
}

}  // namespace gmg
