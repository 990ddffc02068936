// gmg_sgs_phase.hpp -- the SSOR wavefront sweep of gmg_sgs.hpp with the dependent steps taken in turn by FOUR waves.
//
// Reference: LA::MPI::PreconditionSSOR with AdditionalData(0.5), /root/reference/src/step-50.cc:970-973 (same
// arithmetic, same order as gmg_sgs.hpp and oracle/gmg_oracle.c:smoother_apply_inverse; bit-identical results).
//
// One wave pays ~1000 cycles per dependent step although the step's true dependence is short: of the ~14 products a
// row adds, only the few whose column was updated by the PREVIOUS step cannot be formed in advance.  A row's sum is
// therefore cut in three (CSR order is kept):
//     HEAD  everything before its first "late" column (a column the previous step updates): gathered and summed ahead;
//     T1    from the first late column to the last one: gathered, multiplied and added in the dependent phase;
//     T2    what follows the last late column: products formed ahead, only ADDED in the dependent phase.
// Four waves of one workgroup take the steps in turn and meet at s_barrier once per PHASE; wave w owns the steps
// t = w, w + 4, ... and spends four phases on each:
//     phase t - 3   P1    the step's records, LDS region of the wave -> registers
//     phase t - 2   COPY  starts the copy of its next block (step t + 4) global -> the same LDS region (global_load_lds:
//                         no registers, no helper waves; ~45 cycles per KB to issue, which is why it has a phase of its own)
//     phase t - 1   P2    head: gathers + partial sum; T2: gathers + products
//     phase t       CRIT  T1 gathers, T1 multiply-adds, T2 adds, the new y, one LDS store
// so that in every phase one wave is in CRIT and the dependent chain is barrier -> <= L1 gathers -> L1 multiply-adds +
// L2 adds -> one LDS store -> barrier.  A fifth wave touches the record stream ahead of the copies so that they hit
// the L2.  Nobody spins: every wave executes exactly n_steps + 3 barriers per range.
//
// Shapes.  The code of a step is straight-line for its shape (G groups of 8 head slots, L1, L2): 51 shapes x 2
// directions are instantiated, the host picks per CHUNK of 32 steps the cheapest shape that holds every row (measured
// phase costs), a row may give the end of its head / the start of its T2 to T1 to fit.  Changing the shape from one
// turn to the next costs a wave ~400 cycles (another stretch of code): per-step shapes were measured and lost; inside a
// shape the dependent phase still stops at the T1 slots the step really uses (pieces of four, one forward branch).
//
// Measured on the 64 k-atom hierarchy (level 1: 170 516 rows, 92 164 coupled, 4 384 stages; MI355X): 2.47 ms per sweep
// pair against 4.78 ms for the one-wave sweep; per step ~625 cycles (parts as the instrumented variant sees them): forward CRIT ~390, backward ~580 (the late columns
// of a backward row are its nearest upper neighbours, FIRST in CSR order: its T1 is longer), P1 ~600, COPY ~500.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gmg_sgs.hpp"

namespace gmg {

constexpr int kPhRegion = 16384;  // bytes of LDS per compute wave for the block it is reading (a block never exceeds it)
constexpr int kPhMaxRows = 32;    // rows per step
constexpr int kPhMaxEntries = 36; // 8 G + L of a range
constexpr int kPhWaves = 4;       // compute waves
constexpr int kPhThreads = 64 * (kPhWaves + 1);  // waves 0..3 compute, wave 4 prefetches the records into the L2
constexpr int kPhJunk = 512;      // LDS bytes behind the regions: [0, 256) the prefetch wave's copies land in, [256, 264) the hand-over words of gmg_sgs_chain.hpp
constexpr int kPhYSlots = (160 * 1024 - kPhWaves * kPhRegion - kPhJunk) / 8 & ~1;  // doubles of y in LDS: 12256
// a range's shape: 8 g head entries, l1 tail entries gathered in the dependent phase, l2 tail entries whose products are formed ahead
__host__ __device__ constexpr bool ph_shape_ok(int g, int l1, int l2) {
  return g >= 0 && g <= 3 && l1 >= 4 && l1 <= 28 && l1 % 4 == 0 && l2 >= 0 && l2 <= 24 && l2 % 8 == 0 && 8 * g + l1 + l2 <= kPhMaxEntries;
}
__host__ __device__ constexpr int ph_stride(int g, int l) {
  const int s = 32 + 96 * g + 12 * l;  // multiple of 16 (l is a multiple of 4)
  return (s / 16) % 2 ? s : s + 16;    // odd multiple of 16: 16-byte LDS reads of consecutive lanes hit distinct banks
}

// One range = consecutive steps of one sweep direction whose working set fits the LDS.  A step's block: 16-byte
// header {-, bytes of the block four steps on (0: none), its offset in the range's stream, that step's shape key | rows << 8 |
// T1 slots in use << 16}, then one record
// per row:  +0 r  +8 1/a_ii  +16 prefix (backward: the forward sweep's sum)  +24 u32 LDS address of the row's y
//           +28 u32 aux (forward: index, in doubles, of the prefix field of the row's backward record)
//           +32 head values [8 G]   tail values [L]   head LDS addresses u32 [8 G]   tail LDS addresses u32 [L]
// (padding: value +0.0, address = the row's own y).
struct PhRange {
  int64_t stream_off;
  int32_t n_steps, ws_off, n_own, n_ws, backward, G, L;  // L = L1 + L2
  uint32_t blk_tab, L1;               // first entry of the range in the block table
  int32_t own_ci0, own_dir, pad0[2];  // own_dir = +1 / -1: slot k of the rows updated here is ycur[own_ci0 + own_dir * k] (ycur is numbered in sweep order); 0: see ws_ci
  uint32_t pf_lead, pf_step;          // prefetch wave: bytes ahead at phase 0, bytes per phase (multiples of 128)
  uint32_t stream_bytes, pad;
};

struct SgsPhaseArgs {
  const PhRange *ranges;
  const int32_t *block_rng;
  int block0;
  char *stream;
  const uint4 *blk_tab;      // per step {offset in its range's stream, bytes, shape key, -}
  const int32_t *ws_ci, *ci_row;
  double *ycur, *y;
  double omega;
  int y_slots;
  unsigned long long *prof;  // per range {cycles of the phase loop, working-set load, write-back, -} (null: off); same results either way
};

namespace ph {

using sw::lds_ld;
using sw::lds_st;
using sw::u32x4;
using sw::f64x2;

__device__ __forceinline__ void bar() {
  asm volatile("" ::: "memory");       // (the compiler moves no LDS access across a phase boundary)
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): my LDS stores are done before the others are released
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// bytes [0, bytes) of src -> LDS at dst, 1 KB per instruction (may copy up to 1008 bytes beyond: the stream is padded).
// Straight-line: a lone wave pays ~35 cycles for every taken branch, so the loop is unrolled over the at most
// kPhRegion / 1024 pieces and left by one forward branch; uniform base + the lane's 32-bit offset.
__device__ __forceinline__ void copy_to_lds(const char *src, uint32_t dst, uint32_t bytes, int lane) {
  const char *s = src + (uint32_t)lane * 16u;
#pragma unroll
  for (int k = 0; k < kPhRegion / 4096; ++k) {  // four pieces share one address register and one M0 value (immediate offsets)
    if ((uint32_t)k * 4096u >= bytes) break;
    const auto *g = (const __attribute__((address_space(1))) void *)(s + k * 4096);
    auto *l = (__attribute__((address_space(3))) void *)(uintptr_t)(dst + (uint32_t)k * 4096u);
    __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
    if ((uint32_t)k * 4096u + 1024u < bytes) __builtin_amdgcn_global_load_lds(g, l, 16, 1024, 0);
    if ((uint32_t)k * 4096u + 2048u < bytes) __builtin_amdgcn_global_load_lds(g, l, 16, 2048, 0);
    if ((uint32_t)k * 4096u + 3072u < bytes) __builtin_amdgcn_global_load_lds(g, l, 16, 3072, 0);
  }
}

template <int G, int L>
struct Rec {
  double hv[G > 0 ? 8 * G : 1], tv[L];
  uint32_t ha[G > 0 ? 8 * G : 1], ta[L];
  double r, invd, prefix, yold, acc;
  uint32_t my, aux;
  int nrows;
};

// what a wave carries from one of its steps to the next
struct Turn {
  uint32_t nx_off, nx_bytes;  // its next block in the range's stream (bytes 0: none)
  int key;                    // its next step: shape | rows << 8 | T1 slots in use << 16 (so that P1 does not wait for the header)
  unsigned long long c_wait, c_p1, c_copy, c_p2, c_crit, c_bar, m0;
};
__host__ __device__ constexpr int ph_key(int g, int l1, int l2) { return g * 64 + (l1 / 4) * 8 + l2 / 8; }

// One step of shape (G, L1, L2) by one wave: four phases, four barriers.
template <int G, int L1, int L2, bool FWD, bool TIMED>
__device__ __forceinline__ void turn(Turn &T, bool first, const char *base, double *stream_d, uint32_t region, int lane, double omega) {
  constexpr int L = L1 + L2;
  constexpr uint32_t stride = (uint32_t)ph_stride(G, L);
  Rec<G, L> C;
  int l1s = L1;  // T1 slots this step really uses (multiple of 4): the dependent phase stops there
  unsigned long long m1 = 0;
#define PH_T(acc) if constexpr (TIMED) { m1 = __builtin_amdgcn_s_memtime(); T.acc += m1 - T.m0; T.m0 = m1; }
  // ---- P1: records -> registers
  // vmcnt counts this wave's vector-memory operations in issue order (gfx9: loads, LDS copies and stores share the counter):
  // all but the newest one done = the block has arrived; the prefix store of my last step may still be on its way (waiting
  // for it too costs 4 % of the sweep)
  if (FWD && !first) __builtin_amdgcn_s_waitcnt(0x0f71);
  else __builtin_amdgcn_s_waitcnt(0x0f70);                 // vmcnt(0)
  PH_T(c_wait)
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
    PH_T(c_p1)
  }
  bar();
  PH_T(c_bar)
  // ---- COPY: my next block (step t + 4) global -> LDS; it is read three phases on
  if (T.nx_bytes) copy_to_lds(base + T.nx_off, region, T.nx_bytes, lane);
  PH_T(c_copy)
  bar();
  PH_T(c_bar)
  // ---- P2: the head (no column of it is written in this phase or the next) and the products behind the last late column
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
    asm volatile("" : "+v"(acc));  // formed here, not after the barrier (the compiler would sink the chain into CRIT)
    C.acc = acc;
#pragma unroll
    for (int k = 0; k < L2; ++k) {
      double pr = C.tv[L1 + k] * y2[k];
      asm volatile("" : "+v"(pr));
      C.tv[L1 + k] = pr;
    }
  }
  PH_T(c_p2)
  bar();
  PH_T(c_bar)
  // ---- CRIT: from the first late column on
  {
    // (in pieces of four, left by one forward branch: the shape is the chunk's, most steps need fewer slots)
    double yt[L1];
#pragma unroll
    for (int q = 0; q < L1 / 4; ++q) {
      if (q > 0 && 4 * q >= l1s) break;
#pragma unroll
      for (int k = 4 * q; k < 4 * q + 4; ++k) yt[k] = lds_ld<double>(C.ta[k]);
    }
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
  if constexpr (TIMED) { __builtin_amdgcn_s_waitcnt(0xc07f); }
  PH_T(c_crit)
  bar();
  PH_T(c_bar)
#undef PH_T
}

// The steps t = w, w + 4, ... of a range by compute wave w.  Every step has its own shape (its key travels in the header
// of the wave's previous block): one indirect branch per turn, in the phase that only reads records.
template <bool FWD, bool TIMED>
__device__ __forceinline__ void sweep(const PhRange *R, const uint4 *tab, const char *stream, double *stream_d, uint32_t region, int w, int lane, double omega,
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
  for (int i = 0; i < w; ++i) bar();
  int done = w;
  if constexpr (TIMED) T.m0 = __builtin_amdgcn_s_memtime();
  while (t < n) {
#define PH_CASE(g, l1, l2) \
    case ph_key(g, l1, l2): \
      do { \
        turn<g, l1, l2, FWD, TIMED>(T, t == w, base, stream_d, region, lane, omega); \
        t += kPhWaves; done += kPhWaves; \
      } while (t < n && (T.key & 0xff) == ph_key(g, l1, l2));  /* (steps of one shape in a row: no dispatch in between) */ \
      break;
    switch (T.key & 0xff) {
      PH_CASE(0, 4, 0) PH_CASE(0, 4, 8) PH_CASE(0, 4, 16) PH_CASE(0, 4, 24) PH_CASE(0, 8, 0) PH_CASE(0, 8, 8)
      PH_CASE(0, 8, 16) PH_CASE(0, 8, 24) PH_CASE(0, 12, 0) PH_CASE(0, 12, 8) PH_CASE(0, 12, 16) PH_CASE(0, 12, 24)
      PH_CASE(0, 16, 0) PH_CASE(0, 16, 8) PH_CASE(0, 16, 16) PH_CASE(0, 20, 0) PH_CASE(0, 20, 8) PH_CASE(0, 20, 16)
      PH_CASE(0, 24, 0) PH_CASE(0, 24, 8) PH_CASE(0, 28, 0) PH_CASE(0, 28, 8) PH_CASE(1, 4, 0) PH_CASE(1, 4, 8)
      PH_CASE(1, 4, 16) PH_CASE(1, 4, 24) PH_CASE(1, 8, 0) PH_CASE(1, 8, 8) PH_CASE(1, 8, 16) PH_CASE(1, 12, 0)
      PH_CASE(1, 12, 8) PH_CASE(1, 12, 16) PH_CASE(1, 16, 0) PH_CASE(1, 16, 8) PH_CASE(1, 20, 0) PH_CASE(1, 20, 8)
      PH_CASE(1, 24, 0) PH_CASE(1, 28, 0) PH_CASE(2, 4, 0) PH_CASE(2, 4, 8) PH_CASE(2, 4, 16) PH_CASE(2, 8, 0)
      PH_CASE(2, 8, 8) PH_CASE(2, 12, 0) PH_CASE(2, 12, 8) PH_CASE(2, 16, 0) PH_CASE(2, 20, 0) PH_CASE(3, 4, 0)
      PH_CASE(3, 4, 8) PH_CASE(3, 8, 0) PH_CASE(3, 12, 0)
      default:  // (the host builds no other shape)
        for (int i = 0; i < kPhWaves; ++i) bar();
        t += kPhWaves; done += kPhWaves;
        break;
    }
#undef PH_CASE
  }
  for (; done < n + kPhWaves - 1; ++done) bar();
  if (TIMED && tp && w == 0 && lane == 0) { tp[0] = T.c_wait; tp[1] = T.c_p1; tp[2] = T.c_copy; tp[3] = T.c_p2; tp[4] = T.c_crit; tp[5] = T.c_bar; }
}

}  // namespace ph

__global__ __launch_bounds__(kPhThreads) void sgs_phase_kernel(SgsPhaseArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];  // at LDS address 0: [y slots][3 regions][junk]
  double *ylds = reinterpret_cast<double *>(lds);
  const uint32_t ring0 = (uint32_t)a.y_slots * 8u;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r_begin = a.block_rng[a.block0 + blockIdx.x], r_end = a.block_rng[a.block0 + blockIdx.x + 1];
  for (int rg = r_begin; rg < r_end; ++rg) {
    const PhRange *Rp = a.ranges + rg;
    struct { int n_steps, ws_off, n_own, n_ws, backward, G, L; uint32_t pf_lead, pf_step, stream_bytes; int64_t stream_off; } R;
    R.n_steps = Rp->n_steps; R.ws_off = Rp->ws_off; R.n_own = Rp->n_own; R.n_ws = Rp->n_ws; R.backward = Rp->backward; R.G = Rp->G; R.L = Rp->L;
    R.pf_lead = Rp->pf_lead; R.pf_step = Rp->pf_step; R.stream_bytes = Rp->stream_bytes; R.stream_off = Rp->stream_off;
    const int32_t *ws = a.ws_ci + R.ws_off;
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    if (a.prof) t0 = __builtin_amdgcn_s_memtime();
    // working set -> LDS.  The rows updated here are a contiguous piece of ycur (ascending in a forward range, descending in a
    // backward one): coalesced loads, no index list; the rows only read (~900 .. 5 000 of ~12 000) are gathered through ws_ci.
    // Branch-free: a lane beyond the end repeats the last element (same value, same slot), so every load of a pass is issued
    // before the first LDS store and nothing is predicated -- with per-element guards this code was ~500 instructions and a
    // dozen taken branches per 16 elements, 10-20 k cycles per range.  The index loads of the gathered rows are issued
    // first: their round trip hides behind the direct part.
    const int own_dir = Rp->own_dir, own_ci0 = Rp->own_ci0;
    const int n_direct = own_dir ? R.n_own : 0;
    constexpr int kU = 20;  // elements per thread and pass: two passes cover the 12 256 slots (one pass of 39 was measured slower: 35 k cycles)
    {
      const int n_g = R.n_ws - n_direct;  // gathered rows: slots [n_direct, n_ws)
      int gi[kU];
      if (n_g > 0) {
#pragma unroll
        for (int j = 0; j < kU; ++j) gi[j] = ws[n_direct + min(tid + j * kPhThreads, n_g - 1)];
      }
      for (int k0 = 0; k0 < n_direct; k0 += kU * kPhThreads) {
        double v[kU];
#pragma unroll
        for (int j = 0; j < kU; ++j) v[j] = a.ycur[own_ci0 + own_dir * min(k0 + tid + j * kPhThreads, n_direct - 1)];
#pragma unroll
        for (int j = 0; j < kU; ++j) ylds[min(k0 + tid + j * kPhThreads, n_direct - 1)] = v[j];
      }
      for (int k0 = 0; k0 < n_g; k0 += kU * kPhThreads) {
        double v[kU];
        if (k0 > 0) {
#pragma unroll
          for (int j = 0; j < kU; ++j) gi[j] = ws[n_direct + min(k0 + tid + j * kPhThreads, n_g - 1)];
        }
#pragma unroll
        for (int j = 0; j < kU; ++j) v[j] = a.ycur[gi[j]];
#pragma unroll
        for (int j = 0; j < kU; ++j) ylds[n_direct + min(k0 + tid + j * kPhThreads, n_g - 1)] = v[j];
      }
    }
    __syncthreads();
    if (a.prof) t1 = __builtin_amdgcn_s_memtime();
    if (wid < kPhWaves) {
      double *stream_d = reinterpret_cast<double *>(a.stream);
      const uint32_t region = ring0 + (uint32_t)wid * (uint32_t)kPhRegion;
      unsigned long long *tp = a.prof ? a.prof + 12 * (size_t)rg + 4 : nullptr;
      const uint4 *tab = a.blk_tab + Rp->blk_tab;
      if (tp) {  // (the instrumented variant is code of its own: the production sweep carries no timer branches)
        if (R.backward) ph::sweep<false, true>(Rp, tab, a.stream, stream_d, region, wid, lane, a.omega, tp);
        else ph::sweep<true, true>(Rp, tab, a.stream, stream_d, region, wid, lane, a.omega, tp);
      } else {
        if (R.backward) ph::sweep<false, false>(Rp, tab, a.stream, stream_d, region, wid, lane, a.omega, nullptr);
        else ph::sweep<true, false>(Rp, tab, a.stream, stream_d, region, wid, lane, a.omega, nullptr);
      }
    } else {
      // prefetch wave: one 4-byte copy per 128-byte line, pf_step bytes per phase, into the junk area
      const char *base = a.stream + R.stream_off;
      const uint32_t junk = ring0 + (uint32_t)kPhWaves * (uint32_t)kPhRegion;
      uint32_t cur = 0;
      const uint32_t first = min(R.pf_lead, R.stream_bytes);
      for (; cur < first; cur += 8192)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + min(cur + (uint32_t)lane * 128u, R.stream_bytes - 4u)),
                                         (__attribute__((address_space(3))) void *)(uintptr_t)junk, 4, 0, 0);
      for (int p = 0; p < R.n_steps + kPhWaves - 1; ++p) {
        const uint32_t end = min(cur + R.pf_step, R.stream_bytes);
        for (; cur < end; cur += 8192)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + min(cur + (uint32_t)lane * 128u, R.stream_bytes - 4u)),
                                           (__attribute__((address_space(3))) void *)(uintptr_t)junk, 4, 0, 0);
        __builtin_amdgcn_s_barrier();
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): prefix stores, leftover copies
    __syncthreads();
    if (a.prof) t2 = __builtin_amdgcn_s_memtime();
    // LDS -> ycur (and, after the backward sweep, y): the same lanes-repeat-the-last-element form
    for (int k0 = 0; k0 < R.n_own; k0 += kU * kPhThreads) {
      int kc[kU], ci[kU];
      double v[kU];
#pragma unroll
      for (int j = 0; j < kU; ++j) kc[j] = min(k0 + tid + j * kPhThreads, R.n_own - 1);
      if (own_dir) {
#pragma unroll
        for (int j = 0; j < kU; ++j) ci[j] = own_ci0 + own_dir * kc[j];
      } else {
#pragma unroll
        for (int j = 0; j < kU; ++j) ci[j] = ws[kc[j]];
      }
#pragma unroll
      for (int j = 0; j < kU; ++j) v[j] = ylds[kc[j]];
      if (R.backward) {
        int row[kU];
#pragma unroll
        for (int j = 0; j < kU; ++j) row[j] = a.ci_row[ci[j]];
#pragma unroll
        for (int j = 0; j < kU; ++j) a.ycur[ci[j]] = v[j];
#pragma unroll
        for (int j = 0; j < kU; ++j) a.y[row[j]] = v[j];
      } else {
#pragma unroll
        for (int j = 0; j < kU; ++j) a.ycur[ci[j]] = v[j];
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
