// gmg_sgs.hpp -- SSOR smoother (Ifpack point relaxation, symmetric Gauss-Seidel, zero start) as a
// dependency-ordered wavefront sweep with the sweep's slice of y in LDS.
//
// Reference: LA::MPI::PreconditionSSOR with AdditionalData(0.5), /root/reference/src/step-50.cc:970-973;
// restated sequentially in oracle/gmg_oracle.c:smoother_apply_inverse.  The sweep is sequential in the
// local row order; what may run concurrently is fixed by the matrix: row i must see the NEW y_j of every
// coupled j < i and the OLD y_j of every coupled j > i.  "Stages" (longest-path levels of that DAG) are
// computed on the host; any schedule that walks the stages in order and adds each row's products in CSR
// order is bit-identical to the sequential sweep.  On the level matrices of an adaptive hierarchy the DAG
// is deep and thin (64 k atoms, level 1: 47 620 coupled rows in 3 722 stages, ~13 rows per stage), so the
// sweep is a chain of ~7 400 dependent steps per application: latency, not bandwidth, and the design
// minimises the time of ONE step:
//   * one workgroup per SSOR block (1 block = the reference on one rank, B blocks = on B ranks: couplings
//     between blocks are dropped); inside it ONE wave computes, lane = row, up to 64 rows of one stage per
//     step -- no barrier and no flag between dependent steps, only the in-order LDS queue of that wave;
//   * y lives in LDS.  A block whose coupled rows do not fit is cut into RANGES of consecutive steps;
//     a range's working set = the rows it updates + the rows it only reads (on the 64 k level 1: 15 000 +
//     ~900), loaded from / written back to a global copy (ycur) when the range starts / ends;
//   * the records of a step (values, LDS slots of the columns, 1/a_ii, the rhs) are laid out in the order
//     the sweep consumes them; three helper waves copy that stream into a 32 KB LDS ring ahead of the
//     compute wave, so no global-memory latency is ever on the dependent path;
//   * the backward sweep of row i adds, in CSR order, first the columns j < i -- whose y_j still hold the forward
//     values, i.e. exactly the partial sum the forward sweep formed for that row -- so the forward step stores its
//     sum into the backward record of the row and the backward step continues from it with the columns j >= i:
//     half the work, bit-identical (needs ascending columns inside a row; otherwise the generic sweep is used);
//   * stored zeros and columns of other blocks are pruned on the host (x + 0.0 * y = x for finite y), the
//     forward sweep from y = 0 keeps only the columns j < i; rows without any coupling (Dirichlet and
//     refinement-edge rows: half of a level matrix) are closed-form and handled by the pre-pass kernel
//     that also scatters the rhs into the stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gmg {

constexpr int kSwRing = 32768;     // bytes of the record ring in LDS
constexpr int kSwChunk = 4096;     // unit the helper waves copy
constexpr int kSwW = 32;           // most entries per sub-step: a row's sum is formed in groups of 8 products, 1..4 groups per sub-step
constexpr int kSwMaxBlock = 8192;  // bytes of one step's records: 3 blocks + 1 chunk fit the ring (no deadlock)

constexpr int kSwYSlots = 15744;   // doubles of y in LDS (123 KB) by default
constexpr uint32_t kSwSpinLimit = 1u << 22;  // polls (> 0.1 s) after which a waiting wave declares the sweep broken
constexpr int kSwThreads = 256;    // wave 0 computes, waves 1..3 stream
#ifdef GMG_EXPERIMENTS
constexpr bool kSwExperiments = true;   // tools/build_experiments.sh: a library of its own, never the shipped one
#else
constexpr bool kSwExperiments = false;
#endif

// One range = consecutive steps of one sweep direction whose working set fits the LDS.
struct SwRange {
  int64_t stream_off;    // byte offset of the range's records (multiple of kSwChunk)
  int32_t stream_bytes;  // multiple of kSwChunk
  int32_t n_steps;       // sub-steps
  int32_t ws_off;        // first entry of the range in ws_ci
  int32_t n_own;         // slots [0, n_own): rows updated in this range (written back at its end)
  int32_t n_ws;          // slots in use: updated rows + rows only read
  int32_t backward;      // 1: second sweep, its results are final
  int32_t first_raw, first_nrows, groups, pad1;  // groups: g of every sub-step of the range
};


// A step (<= 64 rows of one stage, lane = row) is cut into SUB-STEPS of exactly 8 g entries per row; g (1..4 groups of
// 8) is fixed for a whole range of steps, chosen by the host so that nearly every step is one sub-step (a 27-point
// level operator: g = 2 forward, 3 backward).  The sweep of a range is therefore ONE straight-line loop body --
// gathers, values, the chain of 8 g multiply-adds -- with a single taken branch per sub-step (a lone wave pays a
// refetch for every taken branch: dispatching on the width per sub-step doubled the time of the sweep); the partial
// sum stays in a register from one sub-step to the next and the last sub-step of a step finishes the rows.
// Records of one sub-step: a 16-byte header, then one record per row (lane-major, so that every field is read with
// an immediate offset from the lane's base address, two or four values per LDS instruction):
//   +0  double r        (rewritten by the pre-pass; read by the last sub-step)      +8  double 1 / a_ii
//   +16 double prefix   (backward, first sub-step: the forward sweep's sum over the columns j < i, written by the
//                        forward sweep's last sub-step)
//   +24 uint32 LDS byte address of the row's own y slot
//   +28 uint32 aux      (forward: index, in doubles, of the row's prefix field in the backward records)
//   +32 double a[8 g]   then uint32 LDS byte address of the column's y slot [8 g]
// rows are padded with a = +0.0, column = the row itself; the record stride is 32 + 96 g rounded up to an odd
// multiple of 16 bytes = 96 g + 48 (16-byte LDS reads of consecutive lanes then hit distinct banks).
struct SwStepHdr {
  uint16_t nrows, flags;       // flags: 1 = first sub-step of its step, 2 = last one
  uint16_t next_nrows, pad;
  uint32_t advance;            // bytes to the next header (>= the raw size: records never straddle the ring end)
  uint32_t next_raw;           // raw bytes of the next sub-step's records (0: last of the range)
};
__host__ __device__ constexpr int sw_stride(int g) { return 96 * g + 48; }

struct SgsWaveArgs {
  const SwRange *ranges;
  const int32_t *block_rng;  // n_blocks + 1: ranges of each block (forward ones first)
  int block0;                // first block of this launch (several ranks: each sweeps its share of the blocks)
  char *stream;
  const int32_t *ws_ci;      // working-set lists (compact row ids), own rows first
  const int32_t *ci_row;     // compact id -> level row
  double *ycur;              // latest value of every coupled row
  double *y;                 // level vector (out)
  double omega;
  int y_slots;
  // pre-pass
  const int32_t *row_ci;     // level row -> compact id, -1: row without couplings
  const int32_t *rpos_f, *rpos_b;  // compact id -> index (in doubles) of its rhs field in the stream
  const double *iso_diag;    // level row -> a_ii (rows without couplings; 0 when no diagonal is stored)
  const double *iso_invd;    // level row -> 1 / a_ii
  const double *r;
  int64_t n_rows;
  int *abort_flag;           // host-visible: set when a wave of the sweep gave up waiting for its partner (results invalid)
  int prof_mode;             // PROFILE variant in a -DGMG_EXPERIMENTS build only (timing experiments, wrong results by design): 1 no chain, 5 empty sub-step;
                             // the shipped library compiles them out (kSwExperiments) and gmg_set_option refuses them
  unsigned long long *prof;  // PROFILE variant: per range {cycles of the sweep, of them waiting for the ring, working-set load, write-back}
};

// rhs -> stream, ycur = 0, and the closed form of the two sweeps for rows without couplings:
//   forward  y1 = 0 + (omega (r - 0)) / a_ii ;  backward  y2 = y1 + (omega (r - a_ii y1)) / a_ii
__global__ __launch_bounds__(256) void sgs_wave_prepass_kernel(SgsWaveArgs a) {
  double *sd = reinterpret_cast<double *>(a.stream);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n_rows; i += (int64_t)gridDim.x * 256) {
    const int ci = a.row_ci[i];
    const double ri = a.r[i];
    if (ci < 0) {
      const double invd = a.iso_invd[i];
      const double y1 = 0.0 + (a.omega * (ri - 0.0)) * invd;
      const double acc = 0.0 + a.iso_diag[i] * y1;
      a.y[i] = y1 + (a.omega * (ri - acc)) * invd;
    } else {
      sd[a.rpos_f[ci]] = ri;
      sd[a.rpos_b[ci]] = ri;
      a.ycur[ci] = 0.0;
    }
  }
}

namespace sw {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// The sweep addresses LDS by absolute byte address (y slot addresses are baked into the records): the kernel has
// no static __shared__, so its dynamic LDS starts at 0 (the host checks hipFuncGetAttributes before using the plan).
template <class T>
__device__ __forceinline__ T lds_ld(uint32_t addr) {
  return *reinterpret_cast<const __attribute__((address_space(3))) T *>(addr);
}
template <class T>
__device__ __forceinline__ void lds_st(uint32_t addr, T v) {
  *reinterpret_cast<__attribute__((address_space(3))) T *>(addr) = v;
}
__device__ __forceinline__ uint32_t ld_acq(uint32_t addr) {
  return __hip_atomic_load(reinterpret_cast<__attribute__((address_space(3))) uint32_t *>(addr), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void st_rel(uint32_t addr, uint32_t v) {
  __hip_atomic_store(reinterpret_cast<__attribute__((address_space(3))) uint32_t *>(addr), v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void st_rlx(uint32_t addr, uint32_t v) {
  __hip_atomic_store(reinterpret_cast<__attribute__((address_space(3))) uint32_t *>(addr), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// what the compute wave holds in registers of the sub-step AFTER the one it is computing: the LDS addresses of the
// columns and the prefix sum, so that the y gathers never wait for a ring read.  Always a fresh value (entries beyond
// the sub-step's groups stay undefined): nothing has to be merged or copied between the register sets.
template <int G>
struct NextCols {
  uint32_t addr[8 * G];
  uint32_t my, aux;
  double prefix;
  u32x4 hdr;  // SwStepHdr as loaded
};

// Lanes beyond the sub-step's rows run on the last row's record (all reads stay in range, control flow stays
// wave-uniform); only the stores that finish a row are masked.
// Lanes beyond the sub-step's rows run on the last row's record (all reads stay in range, control flow stays
// wave-uniform); only the stores that finish a row are masked.
template <int G>
__device__ __forceinline__ NextCols<G> load_next(uint32_t blk, int nrows, int lane) {
  NextCols<G> N;
  N.hdr = lds_ld<u32x4>(blk);
  const uint32_t rec = blk + 16 + (uint32_t)min(lane, nrows - 1) * (uint32_t)sw_stride(G);
  const u32x4 q = lds_ld<u32x4>(rec + 16);
  N.prefix = __hiloint2double((int)q.y, (int)q.x);
  N.my = q.z; N.aux = q.w;
#pragma unroll
  for (int j = 0; j < 2 * G; ++j) {
    const u32x4 c = lds_ld<u32x4>(rec + 32 + 64 * G + 16 * j);
    N.addr[4 * j] = c.x; N.addr[4 * j + 1] = c.y; N.addr[4 * j + 2] = c.z; N.addr[4 * j + 3] = c.w;
  }
  return N;
}

// The sweep of one range by the compute wave.  ring0 / ctr0: LDS addresses of the record ring and of the four counters
// ([0..2] chunks copied by helper 0..2, [3] bytes consumed).
// Forward: acc = sum over the columns j < i, kept for the backward sweep; y_i = 0 + (omega (r_i - acc)) / a_ii.
// Backward: acc = that sum, continued over the columns j >= i; y_i += (omega (r_i - acc)) / a_ii.
template <int G, bool FWD, bool PROFILE>
__device__ __forceinline__ void sweep_range(const SwRange &R, uint32_t ring0, uint32_t ctr0, int lane, double omega, double *stream_d,
                                            unsigned long long &t_wait, int mode) {
  uint32_t pos = 0, avail = 0;
  bool ok = true;  // false once a wave of the workgroup gave up waiting (ctr[4]): everybody leaves for the barrier
  auto wait_for = [&](uint32_t end) {
    if (__builtin_expect(avail >= end, 1)) return;  // the helpers run ahead: keep the hot path free of taken branches
    unsigned long long w0 = 0;
    if constexpr (PROFILE) w0 = __builtin_amdgcn_s_memtime();
    for (uint32_t spins = 0; avail < end; ++spins) {
      const uint32_t d0 = ld_acq(ctr0), d1 = ld_acq(ctr0 + 4), d2 = ld_acq(ctr0 + 8);
      const uint32_t q = min(min(d0 * 3u, d1 * 3u + 1u), d2 * 3u + 2u);  // chunks [0, q) are in the ring
      avail = (uint32_t)__builtin_amdgcn_readfirstlane((int)(q * (uint32_t)kSwChunk));
      if (avail >= end) break;
      __builtin_amdgcn_s_sleep(1);
      if (spins > kSwSpinLimit) st_rlx(ctr0 + 16, 1u);
      if ((spins & 63u) == 63u && __builtin_amdgcn_readfirstlane((int)ld_acq(ctr0 + 16))) { ok = false; break; }
    }
    if constexpr (PROFILE) t_wait += __builtin_amdgcn_s_memtime() - w0;
  };
  int nrows = R.first_nrows;
  double carry = 0.0;
  // one sub-step on the register set `cur`; fills `nxt` with the set of the sub-step after it (two sets used alternately:
  // no register copies between sub-steps)
  auto one_step = [&](const NextCols<G> &cur, NextCols<G> &nxt) {
    const uint32_t blk = ring0 + (pos & (uint32_t)(kSwRing - 1));
    // header: {nrows | flags << 16, next_nrows, advance, next_raw}
    const uint32_t flags = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur.hdr.x) >> 16;
    const int n_nrows = __builtin_amdgcn_readfirstlane((int)cur.hdr.y) & 0xffff;
    const uint32_t advance = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur.hdr.z);
    const uint32_t next_raw = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur.hdr.w);
    const uint32_t rec = blk + 16 + (uint32_t)min(lane, nrows - 1) * (uint32_t)sw_stride(G);
    if (!(PROFILE && kSwExperiments && mode == 5)) {
      // ---- every LDS read of the sub-step goes out first: y gathers, own y, values, rhs; then the next sub-step's columns
      double yv[8 * G], av[8 * G];
#pragma unroll
      for (int k = 0; k < 8 * G; ++k) yv[k] = lds_ld<double>(cur.addr[k]);
      double yold = 0.0;
      if constexpr (!FWD) yold = lds_ld<double>(cur.my);
#pragma unroll
      for (int j = 0; j < 4 * G; ++j) {
        const f64x2 a2 = lds_ld<f64x2>(rec + 32 + 16 * j);
        av[2 * j] = a2.x; av[2 * j + 1] = a2.y;
      }
      const f64x2 ri = lds_ld<f64x2>(rec);  // r, 1 / a_ii
      // everything in front of this sub-step's records may be overwritten (relaxed: the LDS queue of this wave is in
      // order, the store cannot overtake the reads issued before it)
      st_rlx(ctr0 + 12, pos);
      if (__builtin_expect(next_raw != 0, 1)) {
        wait_for(pos + advance + next_raw);
        nxt = load_next<G>(ring0 + ((pos + advance) & (uint32_t)(kSwRing - 1)), n_nrows, lane);  // (after a failed wait: stale bytes, LDS only)
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- the dependent chain, in CSR order
      double acc = (flags & 1u) ? (FWD ? 0.0 : cur.prefix) : carry;
      if (!(PROFILE && kSwExperiments) || mode != 1) {
#pragma unroll
        for (int k = 0; k < 8 * G; ++k) acc += av[k] * yv[k];
      }
      carry = acc;
      if ((flags & 2u) && lane < nrows && ok) {  // (ok: no global store from stale records)
        if constexpr (FWD) stream_d[cur.aux] = acc;
        lds_st<double>(cur.my, yold + (omega * (ri.x - acc)) * ri.y);
      }
    } else {
      st_rlx(ctr0 + 12, pos);
      if (next_raw) { wait_for(pos + advance + next_raw); nxt = load_next<G>(ring0 + ((pos + advance) & (uint32_t)(kSwRing - 1)), n_nrows, lane); }
    }
    pos += advance;
    nrows = n_nrows;
  };
  wait_for((uint32_t)R.first_raw);
  if (!ok) return;
  NextCols<G> C0 = load_next<G>(ring0, nrows, lane), C1 = C0;
  for (int s = 0; s < R.n_steps && ok; s += 2) {
    one_step(C0, C1);
    if (s + 1 < R.n_steps) one_step(C1, C0);
  }
}

template <bool FWD, bool PROFILE>
__device__ __forceinline__ void sweep_dispatch(const SwRange &R, uint32_t ring0, uint32_t ctr0, int lane, double omega, double *stream_d,
                                               unsigned long long &t_wait, int mode) {
  switch (R.groups) {
    case 1: sweep_range<1, FWD, PROFILE>(R, ring0, ctr0, lane, omega, stream_d, t_wait, mode); break;
    case 2: sweep_range<2, FWD, PROFILE>(R, ring0, ctr0, lane, omega, stream_d, t_wait, mode); break;
    case 3: sweep_range<3, FWD, PROFILE>(R, ring0, ctr0, lane, omega, stream_d, t_wait, mode); break;
    default: sweep_range<4, FWD, PROFILE>(R, ring0, ctr0, lane, omega, stream_d, t_wait, mode); break;
  }
}

}  // namespace sw

template <bool PROFILE>
__global__ __launch_bounds__(kSwThreads) void sgs_wave_kernel(SgsWaveArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];  // at LDS address 0: [y slots][record ring][5 counters]
  double *ylds = reinterpret_cast<double *>(lds);
  const uint32_t ring0 = (uint32_t)a.y_slots * 8u, ctr0 = ring0 + (uint32_t)kSwRing;
  char *ring = lds + ring0;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r_begin = a.block_rng[a.block0 + blockIdx.x], r_end = a.block_rng[a.block0 + blockIdx.x + 1];
  for (int rg = r_begin; rg < r_end; ++rg) {
    const SwRange R = a.ranges[rg];
    const int32_t *ws = a.ws_ci + R.ws_off;
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t_wait = 0;
    if constexpr (PROFILE) t0 = __builtin_amdgcn_s_memtime();
    for (int k0 = tid; k0 < R.n_ws; k0 += 8 * kSwThreads) {  // eight independent gathers in flight per thread
      int ci[8];
      double v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) ci[j] = k0 + j * kSwThreads < R.n_ws ? ws[k0 + j * kSwThreads] : -1;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = ci[j] >= 0 ? a.ycur[ci[j]] : 0.0;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (ci[j] >= 0) ylds[k0 + j * kSwThreads] = v[j];
    }
    if (tid < 5) sw::st_rlx(ctr0 + 4 * tid, 0u);
    __syncthreads();
    if constexpr (PROFILE) t1 = __builtin_amdgcn_s_memtime();
    if (wid == 0) {
      // ---------------- compute wave
      double *stream_d = reinterpret_cast<double *>(a.stream);
      if (R.backward) sw::sweep_dispatch<false, PROFILE>(R, ring0, ctr0, lane, a.omega, stream_d, t_wait, a.prof_mode);
      else sw::sweep_dispatch<true, PROFILE>(R, ring0, ctr0, lane, a.omega, stream_d, t_wait, a.prof_mode);
    } else {
      // ---------------- helper waves: stream -> ring, one 4 KB chunk at a time, chunk q by helper q % 3
      const int hw = wid - 1;
      const int n_chunks = R.stream_bytes / kSwChunk;
      const uint4 *src = reinterpret_cast<const uint4 *>(a.stream + R.stream_off);
      uint32_t done = 0;
      for (int q = hw; q < n_chunks; q += 3) {
        static_assert(kSwChunk == 4096, "four 1 KB pieces per chunk");
        const uint4 *sq = src + (size_t)q * (kSwChunk / 16) + lane;
        const uint4 b0 = sq[0], b1 = sq[64], b2 = sq[128], b3 = sq[192];
        const uint32_t need = (uint32_t)(q + 1) * (uint32_t)kSwChunk;
        bool h_ok = true;
        for (uint32_t spins = 0; need > (uint32_t)__builtin_amdgcn_readfirstlane((int)sw::ld_acq(ctr0 + 12)) + (uint32_t)kSwRing; ++spins) {
          __builtin_amdgcn_s_sleep(2);
          if (spins > kSwSpinLimit) sw::st_rlx(ctr0 + 16, 1u);
          if ((spins & 63u) == 63u && __builtin_amdgcn_readfirstlane((int)sw::ld_acq(ctr0 + 16))) { h_ok = false; break; }
        }
        if (!h_ok) break;
        uint4 *dst = reinterpret_cast<uint4 *>(ring + (((uint32_t)q * (uint32_t)kSwChunk) & (uint32_t)(kSwRing - 1))) + lane;
        dst[0] = b0; dst[64] = b1; dst[128] = b2; dst[192] = b3;
        ++done;
        sw::st_rel(ctr0 + 4 * hw, done);
      }
    }
    __syncthreads();
    if (sw::ld_acq(ctr0 + 16)) {  // a wave gave up waiting: the sweep is broken (never seen; guards the GPU against a hang)
      if (tid == 0) *a.abort_flag = 1;
      return;
    }
    if constexpr (PROFILE) t2 = __builtin_amdgcn_s_memtime();
    for (int k0 = tid; k0 < R.n_own; k0 += 8 * kSwThreads) {
      int ci[8], row[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) ci[j] = k0 + j * kSwThreads < R.n_own ? ws[k0 + j * kSwThreads] : -1;
#pragma unroll
      for (int j = 0; j < 8; ++j) row[j] = (R.backward && ci[j] >= 0) ? a.ci_row[ci[j]] : -1;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (ci[j] >= 0) {
          const double v = ylds[k0 + j * kSwThreads];
          a.ycur[ci[j]] = v;
          if (row[j] >= 0) a.y[row[j]] = v;
        }
    }
    __syncthreads();
    if constexpr (PROFILE) {
      t3 = __builtin_amdgcn_s_memtime();
      if (tid == 0) {
        unsigned long long *o = a.prof + 4 * (size_t)rg;
        o[0] = t2 - t1; o[1] = t_wait; o[2] = t1 - t0; o[3] = t3 - t2;
      }
    }
  }
}

}  // namespace gmg
