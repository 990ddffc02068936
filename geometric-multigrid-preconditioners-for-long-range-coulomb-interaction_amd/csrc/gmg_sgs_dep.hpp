// gmg_sgs_dep.hpp -- the SSOR wavefront sweep with ONE wave running all dependent steps back to back, fed by three
// preparing waves through a ring of compact records in LDS.
//
// Reference: LA::MPI::PreconditionSSOR with AdditionalData(0.5), /root/reference/src/step-50.cc:970-973 (same arithmetic, same
// order as gmg_sgs_phase.hpp, gmg_sgs.hpp and oracle/gmg_oracle.c:smoother_apply_inverse: bit-identical results).
//
// Why.  In the four-wave sweep (gmg_sgs_phase.hpp) every dependent step ends at an s_barrier: ~250 of its ~625 cycles are
// drain + barrier + restart, and a hand-over through an LDS word costs the same (gmg_sgs_chain.hpp, measured).  The one
// mechanism on this machine that passes a y value from one dependent step to the next for the price of an LDS round trip is
// a single wave's own in-order LDS queue (round 2's one-wave sweep) -- but that wave did everything itself, ~140
// instructions per step at one instruction per 6-7 cycles.  Here the dependent wave (D) does only what depends on the
// previous steps:
//     T1 gathers, T1 multiply-adds, T2 adds, the new y, one LDS store                       (~60 instructions per step)
// and everything else happens ahead of it in three PREP waves that take the steps in turn:
//     raw record (global, field-major: coalesced 16-byte loads straight into registers, no LDS staging, no copy phase)
//     -> head gathers + head sum, T2 gathers + T2 products, the row's old y                  (needs the y of steps <= t - 4)
//     -> compact record {head sum, r, 1/a_ii, old y, slot of y, prefix index, T1 values + slots, T2 products} into slot t % 4
//        of an LDS ring, then the slot's ready word.
// "Late" (T1) therefore means: updated within the last THREE steps (the host cuts the rows accordingly: per-step maxima of
// T1 grow from 2.6 / 1.3 entries to 5.3 / 3.4, forward / backward, on the 64 k-atom level 1; tools/sgs_stats_d.py).
// Synchronisation: `done` (last step D finished; prep of step t waits for done >= t - 4, which also frees ring slot t % 4)
// and the first word of a slot's header (the step whose record the slot holds, written after the record; D looks at it two
// steps ahead, off the dependent chain).  All polls are bounded (abort word -> GMG_ERR_HIP) as in gmg_sgs.hpp.  Four waves,
// one per SIMD.
//
// Status: OPT-IN (option sgs_dep=1), bit-identical to the four-wave sweep, and SLOWER: 4.47 ms against 2.48 ms per level-1
// application at 64 k atoms (gpurun_out/r3f, round 3).  The profile (sgs_phase_profile=1 with sgs_dep=1) has the dependent
// wave waiting 1-4 % of its time and the preparing waves 22-33 % of theirs, at 890 (forward) / 1190 (backward) cycles per
// step: D is bound by its OWN instruction stream.  A window of three steps puts 16 T1 slots into half of the steps, so a
// step is 16 gathers + ~20 16-byte record reads (LDS data alone: ~220 cycles) followed by L1 + L2 + 4 dependent fp64
// operations at ~8 cycles each -- about twice what the barriers cost the four-wave sweep, which spreads the same reads over
// four SIMDs.  Kept as the measured answer to "one wave, no barriers"; the default stays gmg_sgs_phase.hpp.
#pragma once
#include "gmg_sgs_phase.hpp"

namespace gmg {

constexpr int kDpLate = 3;          // a column updated within the last kDpLate steps is gathered by the dependent wave
constexpr int kDpPrep = 3;          // preparing waves
constexpr int kDpSlots = 4;         // ring slots: prep of step t starts when step t - 4 is finished, i.e. when slot t % 4 is free
constexpr int kDpThreads = 64 * (1 + kDpPrep);
constexpr int kDpMaxL1 = 28;        // T1 slots of a compact record (the shapes of gmg_sgs_phase.hpp: 8 g + l1 + l2 <= 36)
constexpr int kDpMaxL2 = 24;
__host__ __device__ constexpr int dp_cstride(int l1, int l2) {  // bytes of a compact record: odd multiple of 16
  const int s = 48 + 12 * l1 + 8 * l2;
  return (s / 16) % 2 ? s : s + 16;
}
// the widest compact record the shapes allow (l1 + l2 <= 36): (28, 8)
constexpr int kDpMaxStride = dp_cstride(28, 8);
static_assert(dp_cstride(12, 24) <= kDpMaxStride && dp_cstride(20, 16) <= kDpMaxStride && dp_cstride(24, 8) <= kDpMaxStride && dp_cstride(16, 16) <= kDpMaxStride, "slot size");
constexpr int kDpSlotBytes = 16 + kPhMaxRows * kDpMaxStride;  // 14 864
constexpr int kDpFlagBytes = 64;   // done, abort, ready[4]
constexpr int kDpYSlots = ((160 * 1024 - kDpSlots * ((kDpSlotBytes + 15) / 16 * 16) - kDpFlagBytes) / 8) & ~1;
constexpr uint32_t kDpSpinLimit = 1u << 21;

namespace dp {

using ph::ph_key;
using sw::lds_ld;
using sw::lds_st;
using sw::u32x4;
using sw::f64x2;

__device__ __forceinline__ int vol_ld32(uint32_t addr) { return *reinterpret_cast<const volatile __attribute__((address_space(3))) int *>(addr); }
__device__ __forceinline__ void vol_st32(uint32_t addr, int v) { *reinterpret_cast<volatile __attribute__((address_space(3))) int *>(addr) = v; }

struct Lds {  // byte addresses
  uint32_t ring, done, abort, ready;
  bool timed;                        // diagnostics (option sgs_phase_profile): cycles spent waiting, per wave
  mutable unsigned long long waited; // (accumulated in registers, written out at the end of the range)
  __device__ uint32_t slot(int t) const { return ring + (uint32_t)(t & (kDpSlots - 1)) * (uint32_t)((kDpSlotBytes + 15) / 16 * 16); }
  __device__ uint32_t ready_of(int t) const { return ready + 4u * (uint32_t)(t & (kDpSlots - 1)); }
};

// wait until *addr >= want (wave-uniform); false: aborted
__device__ __forceinline__ bool wait_ge(uint32_t addr, int want, const Lds &M) {
  unsigned long long t0 = 0;
  if (M.timed) t0 = __builtin_amdgcn_s_memtime();
  for (uint32_t spins = 0;; ++spins) {
    const int d = __builtin_amdgcn_readfirstlane(vol_ld32(addr));
    if (__builtin_expect(d >= want, 1)) {
      if (M.timed) M.waited += __builtin_amdgcn_s_memtime() - t0;
      return true;
    }
    __builtin_amdgcn_s_sleep(1);
    if (spins > kDpSpinLimit) vol_st32(M.abort, 1);
    if ((spins & 63u) == 63u && __builtin_amdgcn_readfirstlane(vol_ld32(M.abort))) return false;
  }
}

// ------------------------------------------------------------------------------------------------ preparing waves
// raw record of one row as the host lays it out (gmg_sgs_phase.hpp: Rec), loaded field by field: unit u of row r sits at
// block + 16 + (u * nrows + r) * 16
template <int G, int L>
struct Raw {
  f64x2 ri;       // r, 1 / a_ii
  u32x4 q;        // prefix (2 words), LDS address of the row's y, aux
  f64x2 hv[G > 0 ? 4 * G : 1], tv[L / 2];
  u32x4 ha[G > 0 ? 2 * G : 1], ta[L / 4];
};
template <int G, int L>
__device__ __forceinline__ void load_raw(Raw<G, L> &R, const char *blk, int nrows, int lane) {
  const char *p = blk + 16 + (size_t)min(lane, nrows - 1) * 16;
  const size_t us = (size_t)nrows * 16;  // bytes between two units
  int u = 0;
  R.ri = *reinterpret_cast<const f64x2 *>(p + us * u++);
  R.q = *reinterpret_cast<const u32x4 *>(p + us * u++);
#pragma unroll
  for (int j = 0; j < 4 * G; ++j) R.hv[j] = *reinterpret_cast<const f64x2 *>(p + us * u++);
#pragma unroll
  for (int j = 0; j < L / 2; ++j) R.tv[j] = *reinterpret_cast<const f64x2 *>(p + us * u++);
#pragma unroll
  for (int j = 0; j < 2 * G; ++j) R.ha[j] = *reinterpret_cast<const u32x4 *>(p + us * u++);
#pragma unroll
  for (int j = 0; j < L / 4; ++j) R.ta[j] = *reinterpret_cast<const u32x4 *>(p + us * u++);
}

// One step t by a preparing wave: its raw record is in R (loaded during the previous turn).  Writes the compact record.
template <int G, int L1, int L2, bool FWD>
__device__ __forceinline__ bool prep_turn(const Raw<G, L1 + L2> &R, int t, int nrows, int l1s, const Lds &M, int lane) {
  if (!wait_ge(M.done, t - kDpSlots, M)) return false;  // the y of steps <= t - 4 is final; slot t % 4 has been consumed
  // ---- head: gathers + sum in CSR order (none of these columns is written by the steps t - 3 .. t - 1)
  double acc = FWD ? 0.0 : __hiloint2double((int)R.q.y, (int)R.q.x);
  if constexpr (G > 0) {
    double yh[8 * G];
#pragma unroll
    for (int j = 0; j < 2 * G; ++j) {
      yh[4 * j] = lds_ld<double>(R.ha[j].x); yh[4 * j + 1] = lds_ld<double>(R.ha[j].y);
      yh[4 * j + 2] = lds_ld<double>(R.ha[j].z); yh[4 * j + 3] = lds_ld<double>(R.ha[j].w);
    }
#pragma unroll
    for (int j = 0; j < 4 * G; ++j) { acc += R.hv[j].x * yh[2 * j]; acc += R.hv[j].y * yh[2 * j + 1]; }
  }
  double yold = 0.0;
  if constexpr (!FWD) yold = lds_ld<double>(R.q.z);
  // ---- T2: products (added by the dependent wave, after T1, in CSR order)
  double pr[L2 > 0 ? L2 : 1];
  if constexpr (L2 > 0) {
    double y2[L2];
#pragma unroll
    for (int j = 0; j < L2 / 4; ++j) {
      const u32x4 a4 = R.ta[L1 / 4 + j];
      y2[4 * j] = lds_ld<double>(a4.x); y2[4 * j + 1] = lds_ld<double>(a4.y); y2[4 * j + 2] = lds_ld<double>(a4.z); y2[4 * j + 3] = lds_ld<double>(a4.w);
    }
#pragma unroll
    for (int j = 0; j < L2 / 2; ++j) { pr[2 * j] = R.tv[L1 / 2 + j].x * y2[2 * j]; pr[2 * j + 1] = R.tv[L1 / 2 + j].y * y2[2 * j + 1]; }
  }
  // ---- compact record -> slot t % 4 (lane-major, stride an odd multiple of 16: conflict-free 16-byte accesses)
  constexpr uint32_t cs = (uint32_t)dp_cstride(L1, L2);
  const uint32_t slot = M.slot(t);
  if (lane < nrows) {
    const uint32_t rec = slot + 16u + (uint32_t)lane * cs;
    lds_st<f64x2>(rec, f64x2{acc, R.ri.x});
    lds_st<f64x2>(rec + 16, f64x2{R.ri.y, yold});
    lds_st<u32x4>(rec + 32, u32x4{R.q.z, R.q.w, 0u, 0u});
#pragma unroll
    for (int j = 0; j < L1 / 2; ++j) lds_st<f64x2>(rec + 48 + 16 * j, R.tv[j]);
#pragma unroll
    for (int j = 0; j < L1 / 4; ++j) lds_st<u32x4>(rec + 48 + 8 * L1 + 16 * j, R.ta[j]);
#pragma unroll
    for (int j = 0; j < L2 / 2; ++j) lds_st<f64x2>(rec + 48 + 12 * L1 + 16 * j, f64x2{pr[2 * j], pr[2 * j + 1]});
  }
  asm volatile("" ::: "memory");
  // the slot's header LAST (behind the record in this wave's LDS queue): its first word is the step it holds = "ready"
  if (lane == 0) *reinterpret_cast<volatile __attribute__((address_space(3))) u32x4 *>(slot) = u32x4{(uint32_t)t, (uint32_t)nrows | ((uint32_t)l1s << 8), (uint32_t)ph_key(0, L1, L2), 0u};
  return true;
}

// The steps t = w, w + 3, ... of a range by preparing wave w.
template <bool FWD>
__device__ __forceinline__ void prep_sweep(const PhRange *Rg, const uint4 *tab, const char *stream, int w, int lane, const Lds &M) {
  const int n = Rg->n_steps;
  const char *base = stream + Rg->stream_off;
  bool ok = true;
  for (int t = w; t < n && ok;) {
    const uint4 e = tab[t];
    const int key = (int)e.z & 0xff;
#define DP_PREP(g, l1, l2) \
    case ph_key(g, l1, l2): { \
      Raw<g, l1 + l2> R; \
      uint4 ee = e; \
      load_raw(R, base + ee.x, (int)(ee.z >> 8) & 0xff, lane); \
      for (;;) { \
        const int nrows = (int)(ee.z >> 8) & 0xff, l1s = (int)(ee.z >> 16); \
        const int tn = t + kDpPrep; \
        uint4 en{0u, 0u, 0xffu, 0u}; \
        if (tn < n) en = tab[tn]; \
        ok = prep_turn<g, l1, l2, FWD>(R, t, nrows, l1s, M, lane); \
        t = tn; \
        if (!ok || t >= n || ((int)en.z & 0xff) != ph_key(g, l1, l2)) break; \
        ee = en; \
        load_raw(R, base + ee.x, (int)(ee.z >> 8) & 0xff, lane);  /* (consumed one turn on: in flight during the wait for `done`) */ \
      } \
    } break;
    switch (key) {
      DP_PREP(0, 4, 0) DP_PREP(0, 4, 8) DP_PREP(0, 4, 16) DP_PREP(0, 4, 24) DP_PREP(0, 8, 0) DP_PREP(0, 8, 8)
      DP_PREP(0, 8, 16) DP_PREP(0, 8, 24) DP_PREP(0, 12, 0) DP_PREP(0, 12, 8) DP_PREP(0, 12, 16) DP_PREP(0, 12, 24)
      DP_PREP(0, 16, 0) DP_PREP(0, 16, 8) DP_PREP(0, 16, 16) DP_PREP(0, 20, 0) DP_PREP(0, 20, 8) DP_PREP(0, 20, 16)
      DP_PREP(0, 24, 0) DP_PREP(0, 24, 8) DP_PREP(0, 28, 0) DP_PREP(0, 28, 8) DP_PREP(1, 4, 0) DP_PREP(1, 4, 8)
      DP_PREP(1, 4, 16) DP_PREP(1, 4, 24) DP_PREP(1, 8, 0) DP_PREP(1, 8, 8) DP_PREP(1, 8, 16) DP_PREP(1, 12, 0)
      DP_PREP(1, 12, 8) DP_PREP(1, 12, 16) DP_PREP(1, 16, 0) DP_PREP(1, 16, 8) DP_PREP(1, 20, 0) DP_PREP(1, 20, 8)
      DP_PREP(1, 24, 0) DP_PREP(1, 28, 0) DP_PREP(2, 4, 0) DP_PREP(2, 4, 8) DP_PREP(2, 4, 16) DP_PREP(2, 8, 0)
      DP_PREP(2, 8, 8) DP_PREP(2, 12, 0) DP_PREP(2, 12, 8) DP_PREP(2, 16, 0) DP_PREP(2, 20, 0) DP_PREP(3, 4, 0)
      DP_PREP(3, 4, 8) DP_PREP(3, 8, 0) DP_PREP(3, 12, 0)
      default:
        vol_st32(M.abort, 1);  // (the host builds no other shape)
        ok = false;
        break;
    }
#undef DP_PREP
  }
}

// ------------------------------------------------------------------------------------------------ the dependent wave
// what the dependent wave holds of a step.  Pipeline (steady state, one shape): during step t it gathers T1(t) with the
// addresses read during step t - 1, reads the VALUES of step t behind the gathers (they do not depend on y), reads the
// ADDRESSES of step t + 1 and looks at the header of step t + 2 -- so that nothing but its own y store stands between the
// dependent chains of consecutive steps.
template <int L1, int L2>
struct Cmp {
  uint32_t my, aux;
  uint32_t ta[L1];
};
struct Hdr { int step, nrows, l1s, key; };
__device__ __forceinline__ Hdr read_hdr(uint32_t slot) {
  const u32x4 h = *reinterpret_cast<const volatile __attribute__((address_space(3))) u32x4 *>(slot);
  Hdr H;
  H.step = __builtin_amdgcn_readfirstlane((int)h.x);
  const int w = __builtin_amdgcn_readfirstlane((int)h.y);
  H.nrows = w & 0xff; H.l1s = w >> 8;
  H.key = __builtin_amdgcn_readfirstlane((int)h.z);
  return H;
}
// the header of step t, waited for (bounded); false: aborted
__device__ __forceinline__ bool wait_hdr(Hdr &H, int t, const Lds &M) {
  unsigned long long t0 = 0;
  if (M.timed) t0 = __builtin_amdgcn_s_memtime();
  for (uint32_t spins = 0;; ++spins) {
    H = read_hdr(M.slot(t));
    if (__builtin_expect(H.step == t, 1)) break;
    __builtin_amdgcn_s_sleep(1);
    if (spins > kDpSpinLimit) vol_st32(M.abort, 1);
    if ((spins & 63u) == 63u && __builtin_amdgcn_readfirstlane(vol_ld32(M.abort))) return false;
  }
  if (M.timed) M.waited += __builtin_amdgcn_s_memtime() - t0;
  return true;
}
template <int L1, int L2>
__device__ __forceinline__ void load_addr(Cmp<L1, L2> &C, uint32_t slot, int nrows, int l1s, int lane) {
  constexpr uint32_t cs = (uint32_t)dp_cstride(L1, L2);
  const uint32_t rec = slot + 16u + (uint32_t)min(lane, nrows - 1) * cs;
  const u32x4 c = lds_ld<u32x4>(rec + 32);
  C.my = c.x; C.aux = c.y;
#pragma unroll
  for (int j = 0; j < L1 / 4; ++j) {
    if (j > 0 && 4 * j >= l1s) break;
    const u32x4 v = lds_ld<u32x4>(rec + 48 + 8 * L1 + 16 * j);
    C.ta[4 * j] = v.x; C.ta[4 * j + 1] = v.y; C.ta[4 * j + 2] = v.z; C.ta[4 * j + 3] = v.w;
  }
}

// The steps of one compact shape, one after the other.  H: the header of step t (ready, this shape).  Returns with t = the
// first step of another shape and H = its header, or t = n.  false: aborted.
template <int L1, int L2, bool FWD>
__device__ __forceinline__ bool dep_run(int &t, int n, Hdr &H, const Lds &M, double *stream_d, double omega, int lane) {
  constexpr uint32_t cs = (uint32_t)dp_cstride(L1, L2);
  Cmp<L1, L2> A, B;
  load_addr(A, M.slot(t), H.nrows, H.l1s, lane);
  Hdr H1{-1, 0, 0, 0};  // header of step t + 1
  if (t + 1 < n && !wait_hdr(H1, t + 1, M)) return false;
  // one step on the addresses in `cur`; leaves the addresses of the next step in `nxt` when it has the same shape
  auto one = [&](Cmp<L1, L2> &cur, Cmp<L1, L2> &nxt, bool &same) -> bool {
    const int nrows = H.nrows, l1s = H.l1s;
    const uint32_t rec = M.slot(t) + 16u + (uint32_t)min(lane, nrows - 1) * cs;
    // ---- T1 gathers: they only wait for this wave's own store of the previous step
    double yt[L1];
#pragma unroll
    for (int q = 0; q < L1 / 4; ++q) {
      if (q > 0 && 4 * q >= l1s) break;
#pragma unroll
      for (int k = 4 * q; k < 4 * q + 4; ++k) yt[k] = lds_ld<double>(cur.ta[k]);
    }
    // ---- the step's values, behind the gathers in the LDS queue (they depend on nothing the previous steps wrote)
    const f64x2 v0 = lds_ld<f64x2>(rec), v1 = lds_ld<f64x2>(rec + 16);
    double tv[L1], pr[L2 > 0 ? L2 : 1];
#pragma unroll
    for (int q = 0; q < L1 / 4; ++q) {
      if (q > 0 && 4 * q >= l1s) break;
      const f64x2 a = lds_ld<f64x2>(rec + 48 + 32 * q), b = lds_ld<f64x2>(rec + 48 + 32 * q + 16);
      tv[4 * q] = a.x; tv[4 * q + 1] = a.y; tv[4 * q + 2] = b.x; tv[4 * q + 3] = b.y;
    }
#pragma unroll
    for (int j = 0; j < L2 / 2; ++j) { const f64x2 v = lds_ld<f64x2>(rec + 48 + 12 * L1 + 16 * j); pr[2 * j] = v.x; pr[2 * j + 1] = v.y; }
    // ---- the addresses of step t + 1 (its header is known) and a look at the header of step t + 2
    same = t + 1 < n && H1.key == ph_key(0, L1, L2);
    if (same) load_addr(nxt, M.slot(t + 1), H1.nrows, H1.l1s, lane);
    Hdr H2{-1, 0, 0, 0};
    if (t + 2 < n) H2 = read_hdr(M.slot(t + 2));
    // ---- the dependent chain, in CSR order: T1 multiply-adds, T2 adds, the new y
    double acc = v0.x;
#pragma unroll
    for (int q = 0; q < L1 / 4; ++q) {
      if (q > 0 && 4 * q >= l1s) break;
#pragma unroll
      for (int k = 4 * q; k < 4 * q + 4; ++k) acc += tv[k] * yt[k];
    }
#pragma unroll
    for (int k = 0; k < L2; ++k) acc += pr[k];
    if (lane < nrows) {
      lds_st<double>(cur.my, v1.y + (omega * (v0.y - acc)) * v1.x);
      if constexpr (FWD) stream_d[cur.aux] = acc;
    }
    asm volatile("" ::: "memory");
    vol_st32(M.done, t);  // (behind the y store in this wave's LDS queue)
    ++t;
    H = H1;
    if (t + 1 < n && H2.step != t + 1) {  // (the preparing waves are normally two steps ahead)
      if (!wait_hdr(H2, t + 1, M)) return false;
    }
    H1 = H2;
    return true;
  };
  for (;;) {
    bool same;
    if (!one(A, B, same)) return false;
    if (t >= n || !same) return true;
    if (!one(B, A, same)) return false;
    if (t >= n || !same) return true;
  }
}

template <bool FWD>
__device__ __forceinline__ void dep_sweep(const PhRange *Rg, const Lds &M, double *stream_d, double omega, int lane) {
  const int n = Rg->n_steps;
  int t = 0;
  bool ok = n > 0;
  Hdr H{-1, 0, 0, 0};
  if (ok) ok = wait_hdr(H, 0, M);
  while (t < n && ok) {
#define DP_DEP(l1, l2) case ph_key(0, l1, l2): ok = dep_run<l1, l2, FWD>(t, n, H, M, stream_d, omega, lane); break;
    switch (H.key) {
      DP_DEP(4, 0) DP_DEP(4, 8) DP_DEP(4, 16) DP_DEP(4, 24) DP_DEP(8, 0) DP_DEP(8, 8) DP_DEP(8, 16) DP_DEP(8, 24)
      DP_DEP(12, 0) DP_DEP(12, 8) DP_DEP(12, 16) DP_DEP(12, 24) DP_DEP(16, 0) DP_DEP(16, 8) DP_DEP(16, 16)
      DP_DEP(20, 0) DP_DEP(20, 8) DP_DEP(20, 16) DP_DEP(24, 0) DP_DEP(24, 8) DP_DEP(28, 0) DP_DEP(28, 8)
      default:
        vol_st32(M.abort, 1);
        ok = false;
        break;
    }
#undef DP_DEP
  }
}

}  // namespace dp

__global__ __launch_bounds__(kDpThreads) void sgs_dep_kernel(SgsPhaseArgs a, int *abort_flag) {
  extern __shared__ __attribute__((aligned(16))) char lds[];  // at LDS address 0: [y slots][4 ring slots][done, abort, ready[4]]
  double *ylds = reinterpret_cast<double *>(lds);
  dp::Lds M;
  M.ring = (uint32_t)a.y_slots * 8u;
  M.done = M.ring + (uint32_t)kDpSlots * (uint32_t)((kDpSlotBytes + 15) / 16 * 16);
  M.abort = M.done + 4u;
  M.ready = M.done + 16u;
  M.timed = a.prof != nullptr;
  M.waited = 0;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r_begin = a.block_rng[a.block0 + blockIdx.x], r_end = a.block_rng[a.block0 + blockIdx.x + 1];
  for (int rg = r_begin; rg < r_end; ++rg) {
    const PhRange *Rp = a.ranges + rg;
    const int n_ws = Rp->n_ws, n_own = Rp->n_own, backward = Rp->backward;
    const int32_t *ws = a.ws_ci + Rp->ws_off;
    for (int k0 = tid; k0 < n_ws; k0 += 16 * kDpThreads) {  // sixteen independent gathers in flight per thread
      int ci[16];
      double v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) ci[j] = k0 + j * kDpThreads < n_ws ? ws[k0 + j * kDpThreads] : -1;
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = ci[j] >= 0 ? a.ycur[ci[j]] : 0.0;
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (ci[j] >= 0) ylds[k0 + j * kDpThreads] = v[j];
    }
    if (tid == 0) {
      dp::vol_st32(M.done, -1); dp::vol_st32(M.abort, 0);
      for (int s = 0; s < kDpSlots; ++s) dp::vol_st32(M.slot(s), -1);  // no slot holds a step yet
    }
    __syncthreads();
    unsigned long long t_begin = 0;
    if (M.timed) { t_begin = __builtin_amdgcn_s_memtime(); M.waited = 0; }
    if (wid == 0) {
      double *stream_d = reinterpret_cast<double *>(a.stream);
      if (backward) dp::dep_sweep<false>(Rp, M, stream_d, a.omega, lane);
      else dp::dep_sweep<true>(Rp, M, stream_d, a.omega, lane);
    } else {
      const uint4 *tab = a.blk_tab + Rp->blk_tab;
      if (backward) dp::prep_sweep<false>(Rp, tab, a.stream, wid - 1, lane, M);
      else dp::prep_sweep<true>(Rp, tab, a.stream, wid - 1, lane, M);
    }
    if (M.timed && lane == 0) {  // per range and wave: {cycles of the sweep, of them waiting}
      a.prof[12 * (size_t)rg + 2 * (size_t)wid] = __builtin_amdgcn_s_memtime() - t_begin;
      a.prof[12 * (size_t)rg + 2 * (size_t)wid + 1] = M.waited;
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): prefix stores
    __syncthreads();
    if (dp::vol_ld32(M.abort)) {  // a wave gave up waiting: the sweep is broken (guards the GPU against a hang)
      if (tid == 0) *abort_flag = 1;
      return;
    }
    for (int k0 = tid; k0 < n_own; k0 += 16 * kDpThreads) {
      int ci[16], row[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) ci[j] = k0 + j * kDpThreads < n_own ? ws[k0 + j * kDpThreads] : -1;
#pragma unroll
      for (int j = 0; j < 16; ++j) row[j] = (backward && ci[j] >= 0) ? a.ci_row[ci[j]] : -1;
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (ci[j] >= 0) {
          const double v = ylds[k0 + j * kDpThreads];
          a.ycur[ci[j]] = v;
          if (row[j] >= 0) a.y[row[j]] = v;
        }
    }
    __syncthreads();
  }
}

}  // namespace gmg
