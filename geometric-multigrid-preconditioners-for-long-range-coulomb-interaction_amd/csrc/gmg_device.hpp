// gmg_device.hpp -- hand-written gfx950 (CDNA4, wave64) kernels of the GMG-CG hot path.
//
// Everything here is HBM/L2-bandwidth work in fp64: no MFMA.  Design rules followed
// (cdna_hip_programming.md G2/G11/G13, MI355X_MICROARCH.md):
//   * CSR SpMV streams val/col with 16-byte-per-lane coalesced loads into an LDS "row window"
//     (products of TILE consecutive nonzeros), then one lane per row sums its segment in CSR
//     order -> bit-identical to the sequential CPU sum, no atomics, deterministic.
//   * workgroup -> tile mapping is XCD-aware: blockIdx%8 labels the XCD (round-robin
//     dispatch), each XCD owns a contiguous eighth of the tiles so the x-vector planes and
//     the matrix slice it touches stay in that XCD's 4 MiB L2 from iteration to iteration.
//   * reductions: per-workgroup partials written in a fixed layout; the *consumer* kernel's
//     workgroups each re-reduce the (<= 1024) partials in a fixed order -- no fp64 atomics,
//     no grid barrier, no host round trip; scalars (alpha, beta, residual) live on the device.
//   * compiled with -ffp-contract=off: a*b+c rounds twice, exactly like the oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace gmg {

constexpr int kThreads = 256;      // 4 waves
constexpr int kTileNnz = 4096;     // LDS row window, doubles (32 KiB) -> 4 workgroups / CU
constexpr int kPasses = kTileNnz / (kThreads * 4);
constexpr int kMaxPartials = 2048; // upper bound of any grid that emits reduction partials

// ---------------------------------------------------------------- reductions

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;  // lane 0 holds the sum
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  return v;
}

// Sum over the workgroup, result broadcast to every thread.  Fixed order => deterministic.
__device__ __forceinline__ double block_sum(double v, double *scratch /* >= 4 doubles of LDS */) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  return ((scratch[0] + scratch[1]) + scratch[2]) + scratch[3];
}

__device__ __forceinline__ double block_max(double v, double *scratch) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  return fmax(fmax(scratch[0], scratch[1]), fmax(scratch[2], scratch[3]));
}

// Every workgroup of a consumer kernel calls this on the producer's partials.
__device__ __forceinline__ double reduce_partials(const double *p, int n, double *scratch) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += kThreads) s += p[i];
  return block_sum(s, scratch);
}

// ---------------------------------------------------------------- coarse-CG device state

constexpr int kXRing = 8;  // iterations whose x += alpha d is applied in one pass (cg_xflush_kernel)

struct CGState {
  double gh[2];       // g.h of the previous / current iteration (slot = iteration parity)
  double res0;        // SolverControl::initial_value()
  double res;         // SolverControl::last_value()
  int it_k1;          // iterations completed, as seen by the SpMV kernel (written by it)
  int it_k2;          // iterations completed, written by the update kernel
  int done;           // 1 once converged / failed: every later kernel returns at once
  int status;         // 0 success, 1 no convergence (step >= max or NaN)
  int iters;          // SolverControl::last_step()
  int pad;
  double alpha[kXRing];  // step lengths of the last kXRing iterations (slot = iteration % kXRing), three-kernel variant
};

// ---------------------------------------------------------------- coarse CG over the peer transport (several ranks)
// With a row-partitioned level 0 an iteration needs the ghost entries of d and two global sums.  Over the peer
// transport (gmg_comm.hpp) none of them is a collective: the direction kernel stores the entries its neighbours need
// straight into THEIR direction vectors and publishes a tag; each sum is ONE one-workgroup kernel (this rank's sum of
// the partials -> every rank's slot -> wait for everybody's tags -> the total, added in rank order: the same bits on
// every rank); a one-workgroup kernel waits for the neighbours' halo tags in front of the SpMV.  The big kernels never
// wait (a grid that fills the GPU and spins would starve the producer when several ranks share one GPU, as in the
// tests).  Everything is indexed by the iteration number modulo 8 and tagged tag0 + iteration: the ranks run in
// lockstep (every iteration needs everybody's sums), so a slot written 8 iterations ago has long been read.
//   area (my mailbox + kPeerCgOffset):  [kind 0 = |g|^2, 1 = d.h][slot 8] { double val[8]; u64 tag[8] }   (2 KB)
//                                       + 2048: halo tags [slot 8][src 8] u64
constexpr int kPeerRanks = 8;
constexpr int kPeerCgOffset = 4096;
struct PeerCG {
  char *area;                   // nullptr: single GPU, or the RCCL path
  char *peer_area[kPeerRanks];  // the same area of every rank (peer_area[me] == area)
  int n_ranks, me;
  unsigned nb_mask;             // ranks whose halo entries I receive
  unsigned long long tag0;      // of this solve
  int *abort_flag;
};
constexpr long long kPeerCgSpinLimit = 30000000LL;  // polls of ~0.5 us with a short sleep: ~15 s (inside a solve the ranks are microseconds apart)

// all ranks' tags of (kind, iteration) have arrived (one thread polls, bounded); false: the exchange is broken
__device__ __forceinline__ bool peer_cg_wait(const PeerCG &pc, const unsigned long long *tags, unsigned mask, unsigned long long want) {
  __shared__ int ok_s;
  if (threadIdx.x == 0) {
    int ok = 1;
    for (int r = 0; r < pc.n_ranks && ok; ++r) {
      if (!((mask >> r) & 1u)) continue;
      // relaxed polls (an acquire load would invalidate the caches on every poll); one acquire fence once all tags are in
      for (long long spins = 0; __hip_atomic_load(tags + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != want; ++spins) {
        __builtin_amdgcn_s_sleep(4);
        if ((spins & 255) == 255 && *(volatile int *)pc.abort_flag) { ok = 0; break; }
        if (spins > kPeerCgSpinLimit) { *pc.abort_flag = 1; ok = 0; break; }
      }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);  // (system scope: the default of the builtin)
    ok_s = ok;
  }
  __syncthreads();
  const bool ok = ok_s != 0;
  __syncthreads();
  return ok;
}
__device__ __forceinline__ const double *peer_cg_vals(const PeerCG &pc, int kind, int it) {
  return reinterpret_cast<const double *>(pc.area + (kind * 8 + (it & 7)) * 128);
}
__device__ __forceinline__ const unsigned long long *peer_cg_tags(const PeerCG &pc, int kind, int it) {
  return reinterpret_cast<const unsigned long long *>(pc.area + (kind * 8 + (it & 7)) * 128 + 64);
}

// ---------------------------------------------------------------- CSR SpMV with LDS row window

enum SpmvMode : int {
  kStore = 0,   // y = acc
  kResid = 1,   // y = b - acc                       (t.sadd(-1,1,defect); defect -= I^T u)
  kAddTo = 2,   // y = b + acc                       (u += P u_c)
  kJacobi = 3,  // y = x_i + (omega (b - acc)) invd  (one damped-Jacobi smoothing step)
  kCheb = 4,    // w = c1 w + omega ((b - acc) invd) ; y = x_i + w   (Chebyshev recurrence step)
};

struct SpmvArgs {
  const int32_t *rowptr;
  const int32_t *col;
  const double *val;
  const int32_t *tile_row;  // n_tiles + 1 row boundaries
  int n_tiles;
  int tiles_per_xcd;
  const double *x;     // input vector (d_old when XFORM)
  double *y;           // output vector (h when XFORM)
  const double *init;  // optional: accumulation starts from init[i] (vmult_add / restrict_and_add)
  const double *b;     // rhs / addend, by mode
  const double *invd;  // 1/a_ii
  double *w;           // Chebyshev direction vector
  double omega, c1;
  // --- XFORM (coarse CG): x_j := beta d_old[j] - g[j] on the fly, d_new and partial d.h out
  const double *g;
  double *dnew;
  CGState *st;
  const double *part_in;  // partials of |g|^2 from the update kernel
  int n_part_in;
  double *part_out;       // partials of d.h, one per workgroup
  double tol;
  int maxit;
};

// SolverCG between two iterations: res = |g| ; SolverControl::check ; beta = gh_new / gh_old.
// Called by every workgroup of the kernel that opens an iteration; all of them derive the
// same scalars from the same partials.  Returns false when the solve is over.
__device__ __forceinline__ bool cg_open_iteration(CGState *st, const double *part_in, int n_part_in, double tol, int maxit,
                                                  double *red, double *beta_out) {
  if (st->done) return false;
  const double gg = reduce_partials(part_in, n_part_in, red);
  const double res = sqrt(gg);
  const int it = st->it_k2;
  const bool conv = res <= tol;
  const bool fail = !conv && (it >= maxit || res != res);
  const bool writer = blockIdx.x == 0 && threadIdx.x == 0;
  if (conv || fail) {
    if (writer) {
      st->done = 1;
      st->status = conv ? 0 : 1;
      st->res = res;
      st->iters = it;
      if (it == 0) st->res0 = res;
    }
    return false;
  }
  const double gh_new = res * res;
  *beta_out = (it == 0) ? 0.0 : gh_new / st->gh[(it & 1) ^ 1];
  if (writer) {
    st->gh[it & 1] = gh_new;
    st->it_k1 = it;
    if (it == 0) st->res0 = res;
  }
  return true;
}

// CG = 0: plain operator application with epilogue MODE
// CG = 1: single-GPU coarse CG, direction update fused in: x_j := beta d_old[j] - g[j]
// CG = 2: distributed coarse CG: x is the already exchanged direction d; emits partial d.h
template <int MODE, int CG>
__global__ __launch_bounds__(kThreads) void spmv_tile_kernel(SpmvArgs a) {
  __shared__ __attribute__((aligned(16))) double prod[kTileNnz + 8];
  __shared__ double red[4];
  const int tid = threadIdx.x;
  constexpr bool XFORM = (CG == 1);

  double beta = 0.0;
  if constexpr (CG == 1) {
    if (!cg_open_iteration(a.st, a.part_in, a.n_part_in, a.tol, a.maxit, red, &beta)) return;
  }
  if constexpr (CG == 2) {
    if (a.st->done) return;
  }

  const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, nb = gridDim.x >> 3;
  const int t_begin = xcd * a.tiles_per_xcd;
  const int t_end = min(t_begin + a.tiles_per_xcd, a.n_tiles);
  double dot_acc = 0.0;

  for (int t = t_begin + lb; t < t_end; t += nb) {
    const int r0 = a.tile_row[t], r1 = a.tile_row[t + 1];
    const int k0 = a.rowptr[r0], k1 = a.rowptr[r1];
    const int ka = k0 & ~3;
    if (k1 - ka <= kTileNnz) {
      // ---- phase A: stream 16 B/lane of col and 2 x 16 B/lane of val, gather x, products -> LDS
      int4 c[kPasses];
      double2 v0[kPasses], v1[kPasses];
#pragma unroll
      for (int p = 0; p < kPasses; ++p) {
        const int base = ka + p * (kThreads * 4) + tid * 4;
        if (base < k1) {
          c[p] = *reinterpret_cast<const int4 *>(a.col + base);
          v0[p] = *reinterpret_cast<const double2 *>(a.val + base);
          v1[p] = *reinterpret_cast<const double2 *>(a.val + base + 2);
        }
      }
#pragma unroll
      for (int p = 0; p < kPasses; ++p) {
        const int base = ka + p * (kThreads * 4) + tid * 4;
        if (base < k1) {
          double x0, x1, x2, x3;
          if constexpr (XFORM) {
            x0 = beta * a.x[c[p].x] - a.g[c[p].x];
            x1 = beta * a.x[c[p].y] - a.g[c[p].y];
            x2 = beta * a.x[c[p].z] - a.g[c[p].z];
            x3 = beta * a.x[c[p].w] - a.g[c[p].w];
          } else {
            x0 = a.x[c[p].x]; x1 = a.x[c[p].y]; x2 = a.x[c[p].z]; x3 = a.x[c[p].w];
          }
          double2 p01, p23;
          p01.x = v0[p].x * x0; p01.y = v0[p].y * x1;
          p23.x = v1[p].x * x2; p23.y = v1[p].y * x3;
          double2 *dst = reinterpret_cast<double2 *>(prod + (base - ka));
          dst[0] = p01;
          dst[1] = p23;
        }
      }
      __syncthreads();
      // ---- phase B: one lane per row, sequential sum in CSR order
      for (int r = r0 + tid; r < r1; r += kThreads) {
        const int s = a.rowptr[r] - ka, e = a.rowptr[r + 1] - ka;
        double acc = a.init ? a.init[r] : 0.0;
        for (int k = s; k < e; ++k) acc += prod[k];
        if constexpr (XFORM) {
          const double dn = beta * a.x[r] - a.g[r];
          a.dnew[r] = dn;
          a.y[r] = acc;
          dot_acc += dn * acc;
        } else if constexpr (CG == 2) {
          a.y[r] = acc;
          dot_acc += a.x[r] * acc;
        } else if constexpr (MODE == kStore) {
          a.y[r] = acc;
        } else if constexpr (MODE == kResid) {
          a.y[r] = a.b[r] - acc;
        } else if constexpr (MODE == kAddTo) {
          a.y[r] = a.b[r] + acc;
        } else if constexpr (MODE == kJacobi) {
          a.y[r] = a.x[r] + (a.omega * (a.b[r] - acc)) * a.invd[r];
        } else if constexpr (MODE == kCheb) {
          const double wn = a.c1 * a.w[r] + a.omega * ((a.b[r] - acc) * a.invd[r]);
          a.w[r] = wn;
          a.y[r] = a.x[r] + wn;
        }
      }
      __syncthreads();
    } else {
      // ---- long row (a single row wider than the window): strided partial sums, fixed order
      double part = 0.0;
      for (int k = k0 + tid; k < k1; k += kThreads) {
        double xv;
        if constexpr (XFORM) xv = beta * a.x[a.col[k]] - a.g[a.col[k]];
        else xv = a.x[a.col[k]];
        part += a.val[k] * xv;
      }
      const double tot = block_sum(part, red);
      if (tid == 0) {
        const int r = r0;
        const double acc = (a.init ? a.init[r] : 0.0) + tot;
        if constexpr (XFORM) {
          const double dn = beta * a.x[r] - a.g[r];
          a.dnew[r] = dn; a.y[r] = acc; dot_acc += dn * acc;
        } else if constexpr (CG == 2) { a.y[r] = acc; dot_acc += a.x[r] * acc; }
        else if constexpr (MODE == kStore) a.y[r] = acc;
        else if constexpr (MODE == kResid) a.y[r] = a.b[r] - acc;
        else if constexpr (MODE == kAddTo) a.y[r] = a.b[r] + acc;
        else if constexpr (MODE == kJacobi) a.y[r] = a.x[r] + (a.omega * (a.b[r] - acc)) * a.invd[r];
        else if constexpr (MODE == kCheb) {
          const double wn = a.c1 * a.w[r] + a.omega * ((a.b[r] - acc) * a.invd[r]);
          a.w[r] = wn; a.y[r] = a.x[r] + wn;
        }
      }
      __syncthreads();
    }
  }
  if constexpr (CG != 0) {
    const double s = block_sum(dot_acc, red);
    if (tid == 0) a.part_out[blockIdx.x] = s;
  }
}

// ---------------------------------------------------------------- SELL-64 SpMV (regular-width operators)
//
// Internal HBM layout for operators whose rows have (nearly) the same length -- the level-0
// lattice and the active-mesh matrix: rows are grouped in slices of 64 (one wavefront), every
// slice is padded to its longest row rounded up to a multiple of 4 and stored "quad-major":
//   col4[q * 64 + lane]           = 4 consecutive column indices of the lane's row   (16 B per lane)
//   val2[(2 q + h) * 64 + lane]   = the matching values, two per 16-B load (h = 0, 1)
// with q running over the quads of slice 0, then slice 1, ...  The matrix therefore streams
// with fully coalesced 16-B-per-lane loads AND, because lane = row, the gathers x[col] of a
// lattice operator are coalesced too (consecutive rows have consecutive columns: ~6 cache
// lines per wave instruction instead of ~24 with the CSR window).  Each lane adds its row's
// products in CSR order (padding = +0.0 at the end), so results stay bit-identical to the
// sequential CPU sum.  A wave owns a CONTIGUOUS run of slices and streams through their quads
// as one software-pipelined loop (two quads = 6 KB per wave prefetched ahead, across slice
// boundaries): no LDS, no barriers, no pipeline refill per slice.  Chosen at upload when the
// padding costs < 12 %.
// Compression of the two streams (decided per operator at upload, values stay bit-exact):
//   VAL8  : the operator has <= 256 distinct values (FE matrices on uniform / 2:1 meshes have a
//           handful) -> one byte per entry indexing a dictionary that the kernel keeps in LDS
//   COL16 : every slice's columns lie within 65536 of the slice's smallest column -> 16-bit
//           offsets from a per-slice base (lattice operators up to ~180^3)
//   PATTERN : slices in which every row has the same (column - row) at every entry position --
//           all the regular part of a lattice; rows that lack a neighbour (domain boundary) get
//           an explicit +0.0 entry there -- store no columns at all: the kernel forms
//           row + delta_j from a tiny per-pattern table (wave-uniform scalar loads)
// so the level-0 lattice streams 1-3 bytes per nonzero instead of 12.
struct SellArgs {
  const int32_t *qptr;   // n_slices + 1, in quads
  const int32_t *sbase;  // COL16: smallest column of each slice
  const void *vals;      // VAL8: uchar4[quads * 64]        else double2[2 * quads * 64]
  const void *cols;      // COL16: ushort4[quads * 64]      else int4[quads * 64]
  const double *dict;    // VAL8: 256 doubles
  const int32_t *spat;   // per slice: column-pattern id, -1 = columns come from the stream (may be null)
  const int32_t *pat;    // [n_patterns][32]: column - row of every entry position of a pattern slice
  int n_slices;
  int n_rows;
  SpmvArgs a;  // vectors, epilogue operands, CG state (rowptr/col/val unused)
};

template <bool VAL8>
struct SellVals {  // the 4 values of one quad as loaded (codes or doubles)
  double2 v0, v1;
};
template <>
struct SellVals<true> {
  uchar4 code;
};

template <int MODE, int CG, bool VAL8, bool COL16>
__global__ __launch_bounds__(kThreads) void spmv_sell_kernel(SellArgs sa) {
  __shared__ double red[4];
  __shared__ double dict[VAL8 ? 256 : 1];
  const SpmvArgs &a = sa.a;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  constexpr bool XFORM = (CG == 1);
  double beta = 0.0;
  if constexpr (CG == 1) {
    if (!cg_open_iteration(a.st, a.part_in, a.n_part_in, a.tol, a.maxit, red, &beta)) return;
  }
  if constexpr (CG == 2) {
    if (a.st->done) return;
  }
  if constexpr (VAL8) {
    dict[threadIdx.x] = sa.dict[threadIdx.x];
    __syncthreads();
  }
  // XCD-aware contiguous ownership: XCD -> eighth of the slices, wave -> contiguous run in it
  const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, nb = gridDim.x >> 3;
  const int per_xcd = (sa.n_slices + 7) >> 3;
  const int x0 = xcd * per_xcd, x1 = min(x0 + per_xcd, sa.n_slices);
  const int waves = nb * 4;
  const int per_wave = (max(x1 - x0, 0) + waves - 1) / waves;
  const int s0 = __builtin_amdgcn_readfirstlane(x0 + (lb * 4 + wid) * per_wave);
  const int s1 = __builtin_amdgcn_readfirstlane(min(s0 + per_wave, x1));
  double dot_acc = 0.0;

  // gathers go through a buffer descriptor: 32-bit element offsets instead of 64-bit address
  // arithmetic per lane (the compressed kernel is VALU / TA-issue bound, not HBM bound)
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(a.x), 0, 0x7FFFFFFF, 0x00020000);
  auto X = [&](int c) -> double {
    const auto raw = __builtin_amdgcn_raw_buffer_load_b64(rx, c << 3, 0, 0);
    const double xv = __builtin_bit_cast(double, raw);
    if constexpr (XFORM) return beta * xv - a.g[c];
    else return xv;
  };

  if (s0 < s1) {
    const int Q1 = sa.qptr[s1];
    int q = sa.qptr[s0];
    int s = s0;
    int qe = sa.qptr[s0 + 1];  // end of the current slice
    int base = COL16 ? sa.sbase[s0] : 0;
    // column-pattern bookkeeping (all wave-uniform): pattern of the current / next slice, first
    // quad of the current slice, end of the next slice (to classify the quad prefetched 2 ahead)
    int qb = q;
    int pid = sa.spat ? sa.spat[s0] : -1;
    int pid_next = (sa.spat && s0 + 1 < s1) ? sa.spat[s0 + 1] : -1;
    int qe_next = (s0 + 1 < s1) ? sa.qptr[s0 + 2] : Q1;
    double acc = (a.init && s * 64 + lane < sa.n_rows) ? a.init[s * 64 + lane] : 0.0;
    using ColT = typename std::conditional<COL16, ushort4, int4>::type;
    const ColT *cbase = reinterpret_cast<const ColT *>(sa.cols) + lane;
    const double2 *vbase = reinterpret_cast<const double2 *>(sa.vals) + lane;
    const uchar4 *kbase = reinterpret_cast<const uchar4 *>(sa.vals) + lane;
    SellVals<VAL8> va{}, vb{}, vc{};
    ColT ca{}, cb{}, cc{};

    auto load_quad = [&](SellVals<VAL8> &V, ColT &C, int Q) {
      // pattern slices need no column stream; a quad beyond the next slice is loaded to be safe
      const bool patterned = Q < qe ? pid >= 0 : (Q < qe_next ? pid_next >= 0 : false);
      if (!patterned) C = cbase[(size_t)Q * 64];
      if constexpr (VAL8) V.code = kbase[(size_t)Q * 64];
      else { V.v0 = vbase[(size_t)(2 * Q) * 64]; V.v1 = vbase[(size_t)(2 * Q + 1) * 64]; }
    };
    auto finish_slice = [&]() {
      const int r = s * 64 + lane;
      if (r < sa.n_rows) {
        if constexpr (XFORM) {
          const double dn = beta * a.x[r] - a.g[r];
          a.dnew[r] = dn; a.y[r] = acc; dot_acc += dn * acc;
        } else if constexpr (CG == 2) {
          a.y[r] = acc; dot_acc += a.x[r] * acc;
        } else if constexpr (MODE == kStore) a.y[r] = acc;
        else if constexpr (MODE == kResid) a.y[r] = a.b[r] - acc;
        else if constexpr (MODE == kAddTo) a.y[r] = a.b[r] + acc;
        else if constexpr (MODE == kJacobi) a.y[r] = a.x[r] + (a.omega * (a.b[r] - acc)) * a.invd[r];
        else if constexpr (MODE == kCheb) {
          const double wn = a.c1 * a.w[r] + a.omega * ((a.b[r] - acc) * a.invd[r]);
          a.w[r] = wn; a.y[r] = a.x[r] + wn;
        }
      }
      ++s;
      if (s < s1) {
        qb = qe;
        qe = sa.qptr[s + 1];
        pid = pid_next;
        pid_next = (sa.spat && s + 1 < s1) ? sa.spat[s + 1] : -1;
        qe_next = (s + 1 < s1) ? sa.qptr[s + 2] : Q1;
        if constexpr (COL16) base = sa.sbase[s];
        acc = (a.init && s * 64 + lane < sa.n_rows) ? a.init[s * 64 + lane] : 0.0;
      }
    };
    // gathers of the current quad first, THEN the prefetch two quads ahead: the in-order vmcnt
    // wait for the gathers leaves the younger stream loads in flight
    auto step = [&](SellVals<VAL8> &V, ColT &C, SellVals<VAL8> &NV, ColT &NC) {
      int c0, c1, c2, c3;
      if (pid >= 0) {
        const int4 dl = *reinterpret_cast<const int4 *>(sa.pat + (size_t)pid * 32 + 4 * (q - qb));
        const int row = s * 64 + lane;
        c0 = row + dl.x; c1 = row + dl.y; c2 = row + dl.z; c3 = row + dl.w;
      } else {
        c0 = base + (int)C.x; c1 = base + (int)C.y; c2 = base + (int)C.z; c3 = base + (int)C.w;
      }
      const double x0_ = X(c0), x1_ = X(c1), x2_ = X(c2), x3_ = X(c3);
      double w0, w1, w2, w3;
      if constexpr (VAL8) { w0 = dict[V.code.x]; w1 = dict[V.code.y]; w2 = dict[V.code.z]; w3 = dict[V.code.w]; }
      else { w0 = V.v0.x; w1 = V.v0.y; w2 = V.v1.x; w3 = V.v1.y; }
      if (q + 2 < Q1) load_quad(NV, NC, q + 2);
      acc += w0 * x0_; acc += w1 * x1_; acc += w2 * x2_; acc += w3 * x3_;
      ++q;
      while (q == qe && s < s1) finish_slice();
    };
    if (q < Q1) load_quad(va, ca, q);  // (a range of empty slices streams nothing)
    if (q + 1 < Q1) load_quad(vb, cb, q + 1);
    while (q == qe && s < s1) finish_slice();  // (empty slices cannot occur: every row has a diagonal)
    while (q < Q1) {
      step(va, ca, vc, cc);
      if (q >= Q1) break;
      step(vb, cb, va, ca);
      if (q >= Q1) break;
      step(vc, cc, vb, cb);
    }
  }
  if constexpr (CG != 0) {
    const double sblock = block_sum(dot_acc, red);
    if (threadIdx.x == 0) a.part_out[blockIdx.x] = sblock;
  }
}

// ---------------------------------------------------------------- SELL-64 pattern kernel (lattice fast path)
//
// The texture-address unit of a CU retires one vector load instruction per ~16 cycles whatever
// its width or the number of active lanes (tools/micro/ta_probe.hip), so the per-entry gathers of
// the kernel above cost 27 x 16 cycles per slice.  In a pattern slice lane L needs x[r0 + L + delta_j];
// for the 27-point lattice row the deltas come as nine runs (x-1, x, x+1).  This kernel reads one
// 16-byte pair {x[c], x[c+1]} per lane and run (c = row + centre offset), gets x[c-1] from the
// neighbouring lane with a DPP wave shift, and fetches the one element left of the wave with a
// scalar load: nine vector loads per slice instead of 27.  Products are still added in CSR order
// with the exact dictionary values, so the result is bit-identical.  Slices with any other pattern
// (the boundary of the numbering) take a plain streamed loop.
constexpr int kSellpMaxClasses = 96;  // row classes (27 coefficients each) the pattern-run kernel keeps in LDS: 20 KB, 6 workgroups / CU

struct SellPatArgs {
  SellArgs sa;
  const uint8_t *rowcls;  // RC: class of every row of a run-pattern slice
  const double *ctab;     // RC: [n_classes][27] coefficients in entry order (exact copies of the dictionary values)
  int n_classes;
  const int32_t *wave_ptr;  // [gridDim.x * 4 + 1]: slice range of every wave (XCD-major, balanced by slice cost on the host)
  const int4 *wave_rr;      // strided fast waves: {first slice, stride, pairs, one more slice or -1}
  int pid0;        // the pattern served by the fast path
  int centre[9];   // column offset (relative to the row) of the centre entry of each of its nine runs
  int col16;       // layout of the column stream used by the other slices
};

// lane L <- value of lane L-1; lane 0 <- its own lane of edge
__device__ __forceinline__ double shift_in_from_left(double v, double edge) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// lane L <- value of lane L+1; lane 63 <- its own lane of edge
__device__ __forceinline__ double shift_in_from_right(double v, double edge) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// lane L <- lane L+1 (lane 63 <- lane 0) / lane L <- lane L-1 (lane 0 <- lane 63)
__device__ __forceinline__ double wave_rol1(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x134 /* wave_rol:1 */, 0xf, 0xf, false);  // (every lane is written)
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x134, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_ror1(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x13c /* wave_ror:1 */, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x13c, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double lane_double(double v, int l) {  // l: wave-uniform
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

constexpr int kSellpWaves = 4;  // resident waves per SIMD (= workgroups per CU) the register budget of the kernel is set for
constexpr int kSellpStrided = 0x20000000;   // flag in wave_ptr[w]: the wave's pairs of slices are described by wave_rr[w]
constexpr int kSellpFastWave = 0x40000000;  // flag in wave_ptr[w]: every slice of wave w is a run-pattern slice with row classes

// RC (row classes): on a lattice the rows of the run-pattern slices have few distinct coefficient
// vectors (interior, next to a boundary face / edge / corner, Dirichlet rows).  Instead of one 8-bit code per ENTRY
// (7 four-byte loads per slice and lane, a bit-field extract + a dictionary read per entry) the kernel reads one
// class byte per ROW and takes the 27 coefficients from an LDS table at immediate offsets.  Same values, same order.
//
// What bounds the kernel at a lattice that fits the caches is the number of vector-memory instructions (a CU's
// address unit takes ~16 cycles per instruction whatever its width) and of VALU instructions that move neighbours
// between lanes.  With row classes two consecutive run-pattern slices are therefore processed as ONE unit of 128
// rows, lane l <-> rows 2l and 2l + 1: per run one 16-byte load per lane brings both rows' middle columns, the left
// column of the even row and the right column of the odd row come from the neighbouring lanes (one DPP shift each),
// and y leaves as one 16-byte store: half the memory instructions and half the shifts per row.  A wave whose slices
// are all run-pattern slices (flagged by the host) needs no slice metadata: the loads of its first unit go out right
// after the scalar load of its slice range, before the tables are in LDS.
template <int MODE, int CG, bool RC>
__global__ __launch_bounds__(kThreads, (CG == 1 ? 3 : kSellpWaves)) void spmv_sellp_kernel(SellPatArgs pa) {
  constexpr bool XFORM = (CG == 1);  // fused opener: the operand is beta d_old - g, formed on the fly (small level 0)
  __shared__ double red[4];
  __shared__ double dict[256];
  __shared__ double ctab[RC ? kSellpMaxClasses * 27 : 1];
  const SellArgs &sa = pa.sa;
  const SpmvArgs &a = sa.a;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  // consecutive workgroups go to consecutive XCDs: wave index = (xcd, workgroup within the xcd, wave), so
  // that every XCD (own L2) walks one contiguous range of slices
  const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, nb = gridDim.x >> 3;
  const int w = (xcd * nb + lb) * 4 + wid;
  const int e0 = __builtin_amdgcn_readfirstlane(pa.wave_ptr[w]);
  const int s0 = e0 & (kSellpStrided - 1);
  const int s1 = __builtin_amdgcn_readfirstlane(pa.wave_ptr[w + 1]) & (kSellpStrided - 1);  // s1 - s0 <= 64 (host)
  const bool fast = RC && (e0 & kSellpFastWave) != 0;
  const uchar4 *kbase = reinterpret_cast<const uchar4 *>(sa.vals) + lane;
  const double *xb[9], *gb[9];
  // lane u <-> the element left of run u's first row, lane 63 - u <-> the element right of its last row (the other
  // lanes repeat lane 0: harmless, in range).  Run after run the register is rotated by one lane to the left (right),
  // so that lane 0 (63) holds the edge the shift of that run takes in.
  int my_edge = pa.centre[0] - 1;
#pragma unroll
  for (int u = 0; u < 9; ++u) {
    xb[u] = a.x + pa.centre[u];
    gb[u] = XFORM ? a.g + pa.centre[u] : nullptr;
    if (lane == u) my_edge = pa.centre[u] - 1;
    if (lane == 63 - u) my_edge = pa.centre[u];
  }
  const bool right_edge = lane >= 55;
  double beta = 0.0;
  // ---- a unit of 128 rows (two slices), lane <-> rows 2 lane, 2 lane + 1
  struct XPair {
    double2 p[9];
    double edge;
    int cls;  // two class bytes
  };
  auto issue_pair = [&](int s, XPair &R) {
    const size_t row = (size_t)s * 64 + 2 * lane;
    // uniform base + 32-bit byte offset of the lane: one address register for the nine loads (n_rows < 2^28)
    const uint32_t off = ((uint32_t)s * 64u + 2u * (uint32_t)lane) * 8u;
#pragma unroll
    for (int u = 0; u < 9; ++u)
      R.p[u] = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(xb[u]) + off);  // 8-byte aligned 16-byte load
    const size_t ec = (size_t)s * 64 + my_edge + (right_edge ? 128 : 0);
    R.edge = a.x[ec];
    if constexpr (RC) R.cls = *reinterpret_cast<const unsigned short *>(pa.rowcls + row);
    if constexpr (XFORM) {
#pragma unroll
      for (int u = 0; u < 9; ++u) {
        const double2 gv = *reinterpret_cast<const double2 *>(gb[u] + row);
        R.p[u].x = beta * R.p[u].x - gv.x;
        R.p[u].y = beta * R.p[u].y - gv.y;
      }
      R.edge = beta * R.edge - a.g[ec];
    }
  };
  // ---- a single run-pattern slice, lane <-> row
  struct XRun {
    double m[9];
    double edge;
    int cls;
  };
  auto issue = [&](int s, XRun &R) {
    const size_t row = (size_t)s * 64 + lane;
#pragma unroll
    for (int u = 0; u < 9; ++u) R.m[u] = xb[u][row];
    const size_t ec = (size_t)s * 64 + my_edge + (right_edge ? 64 : 0);
    R.edge = a.x[ec];
    if constexpr (RC) R.cls = pa.rowcls[row];
    if constexpr (XFORM) {
#pragma unroll
      for (int u = 0; u < 9; ++u) R.m[u] = beta * R.m[u] - gb[u][row];
      R.edge = beta * R.edge - a.g[ec];
    }
  };
  XPair A;
  bool a_ready = false;
  // slice metadata of the whole wave in one round trip, together with the dictionary: lane i <-> slice s0 + i
  int m_qb = 0, m_qe = 0, m_pid = -1;
  // a fast wave's pairs of slices: rr_first + k rr_stride, k < rr_n, then possibly one more slice
  int rr_first = s0, rr_stride = 2, rr_n = (s1 - s0) >> 1, rr_extra = ((s1 - s0) & 1) ? s1 - 1 : -1;
  if (fast) {
    m_pid = pa.pid0;
    if (e0 & kSellpStrided) {
      const int4 d = pa.wave_rr[w];
      rr_first = __builtin_amdgcn_readfirstlane(d.x); rr_stride = __builtin_amdgcn_readfirstlane(d.y);
      rr_n = __builtin_amdgcn_readfirstlane(d.z); rr_extra = __builtin_amdgcn_readfirstlane(d.w);
    }
    if (!XFORM && rr_n > 0) { issue_pair(rr_first, A); a_ready = true; }
  } else if (s0 + lane < s1) {
    m_qb = sa.qptr[s0 + lane];
    m_qe = sa.qptr[s0 + lane + 1];
    m_pid = sa.spat[s0 + lane];
  }
  const double dict_mine = sa.dict[threadIdx.x];
  if constexpr (CG == 1) {
    if (!cg_open_iteration(a.st, a.part_in, a.n_part_in, a.tol, a.maxit, red, &beta)) return;
  }
  if constexpr (CG == 2) {
    if (a.st->done) return;  // read after the loads above were issued: one round trip for all of them
  }
  dict[threadIdx.x] = dict_mine;
  if constexpr (RC) {
    for (int i = threadIdx.x; i < pa.n_classes * 27; i += kThreads) ctab[i] = pa.ctab[i];
  }
  __syncthreads();
  auto X = [&](size_t c) -> double {
    if constexpr (XFORM) return beta * a.x[c] - a.g[c];
    else return a.x[c];
  };
  double dot_acc = 0.0;
  // what happens to a finished row sum
  auto finish = [&](int r, double acc, double self, bool have_self) {
    if constexpr (CG != 0) {
      if (!have_self) self = X((size_t)r);
      if constexpr (XFORM) a.dnew[r] = self;
      a.y[r] = acc; dot_acc += self * acc;
    } else if constexpr (MODE == kStore) a.y[r] = acc;
    else if constexpr (MODE == kResid) a.y[r] = a.b[r] - acc;
    else if constexpr (MODE == kAddTo) a.y[r] = a.b[r] + acc;
    else if constexpr (MODE == kJacobi) a.y[r] = a.x[r] + (a.omega * (a.b[r] - acc)) * a.invd[r];
    else if constexpr (MODE == kCheb) {
      const double wn = a.c1 * a.w[r] + a.omega * ((a.b[r] - acc) * a.invd[r]);
      a.w[r] = wn; a.y[r] = a.x[r] + wn;
    }
  };
  auto ld2 = [&](const double *q, size_t r) -> double2 { return *reinterpret_cast<const double2 *>(q + r); };
  auto st2 = [&](double *q, size_t r, double v0, double v1) { *reinterpret_cast<double2 *>(q + r) = double2{v0, v1}; };

  // ---- two run-pattern slices as one unit (all 128 rows exist)
  auto sum_pair = [&](const int s, const XPair &A) {
    {
      const size_t row = (size_t)s * 64 + 2 * lane;  // even: 16-byte aligned vectors
      double acc0 = 0.0, acc1 = 0.0;
      if (a.init) { const double2 iv = ld2(a.init, row); acc0 = iv.x; acc1 = iv.y; }
      // The coefficients of run u + 1 are read from the LDS table while run u is summed, and no earlier: 54 of them
      // do not fit the registers, and left alone the compiler reads all of them first and spills.  Their LDS addresses
      // are therefore made to "depend" on the sums as they stand when run u starts (an empty asm).
      typedef const __attribute__((address_space(3))) double lds_cdouble;
      const uint32_t tab = (uint32_t)(uintptr_t)(lds_cdouble *)ctab;
      uint32_t b0 = tab + (uint32_t)(A.cls & 0xff) * 216u, b1 = tab + (uint32_t)((A.cls >> 8) & 0xff) * 216u;
      auto coef = [](uint32_t base, int j) -> double { return *reinterpret_cast<lds_cdouble *>((uintptr_t)(base + 8u * (uint32_t)j)); };
      double c0[3] = {coef(b0, 0), coef(b0, 1), coef(b0, 2)}, c1[3] = {coef(b1, 0), coef(b1, 1), coef(b1, 2)};
      double el = A.edge, er = A.edge;
      double t0 = acc0, t1 = acc1;  // the sums one run back
#pragma unroll
      for (int u = 0; u < 9; ++u) {
        double n0[3] = {0.0, 0.0, 0.0}, n1[3] = {0.0, 0.0, 0.0};
        if (u < 8) {
          asm volatile("" : "+v"(b0), "+v"(b1) : "v"(t0), "v"(t1));
          t0 = acc0; t1 = acc1;
#pragma unroll
          for (int j = 0; j < 3; ++j) { n0[j] = coef(b0, 3 * u + 3 + j); n1[j] = coef(b1, 3 * u + 3 + j); }
        }
        const double left = shift_in_from_left(A.p[u].y, el);
        const double right = shift_in_from_right(A.p[u].x, er);
        if (u < 8) { el = wave_rol1(el); er = wave_ror1(er); }
        acc0 += c0[0] * left;
        acc0 += c0[1] * A.p[u].x;
        acc0 += c0[2] * A.p[u].y;
        acc1 += c1[0] * A.p[u].x;
        acc1 += c1[1] * A.p[u].y;
        acc1 += c1[2] * right;
#pragma unroll
        for (int j = 0; j < 3; ++j) { c0[j] = n0[j]; c1[j] = n1[j]; }
      }
      if constexpr (CG != 0) {
        double2 self;
        if (pa.centre[4] == 0) self = A.p[4];
        else { self.x = X(row); self.y = X(row + 1); }
        if constexpr (XFORM) st2(a.dnew, row, self.x, self.y);
        st2(a.y, row, acc0, acc1);
        dot_acc += self.x * acc0;
        dot_acc += self.y * acc1;
      } else if constexpr (MODE == kStore) st2(a.y, row, acc0, acc1);
      else if constexpr (MODE == kResid) { const double2 bv = ld2(a.b, row); st2(a.y, row, bv.x - acc0, bv.y - acc1); }
      else if constexpr (MODE == kAddTo) { const double2 bv = ld2(a.b, row); st2(a.y, row, bv.x + acc0, bv.y + acc1); }
      else if constexpr (MODE == kJacobi) {
        const double2 bv = ld2(a.b, row), xv = ld2(a.x, row), dv = ld2(a.invd, row);
        st2(a.y, row, xv.x + (a.omega * (bv.x - acc0)) * dv.x, xv.y + (a.omega * (bv.y - acc1)) * dv.y);
      } else if constexpr (MODE == kCheb) {
        const double2 bv = ld2(a.b, row), xv = ld2(a.x, row), dv = ld2(a.invd, row), wv = ld2(a.w, row);
        const double wn0 = a.c1 * wv.x + a.omega * ((bv.x - acc0) * dv.x), wn1 = a.c1 * wv.y + a.omega * ((bv.y - acc1) * dv.y);
        st2(a.w, row, wn0, wn1);
        st2(a.y, row, xv.x + wn0, xv.y + wn1);
      }
    }
  };
  int s_begin = s0, s_end = s1;
  if (fast) {
    // nothing but pairs (and at most one slice after them): the first pair's loads are already in flight
    XPair P = A;
    for (int k = 0; k < rr_n; ++k) {
      const int sp = rr_first + k * rr_stride;
      if (!a_ready) issue_pair(sp, P);
      a_ready = false;
      sum_pair(sp, P);
    }
    s_begin = rr_extra >= 0 ? rr_extra : 0;
    s_end = rr_extra >= 0 ? rr_extra + 1 : 0;
  }
  for (int s = s_begin; s < s_end;) {
    const int i = fast ? 0 : s - s0;  // (a fast wave holds the same metadata in every lane)
    const int pid = __builtin_amdgcn_readlane(m_pid, i);
    if (RC && !fast && pid == pa.pid0 && s + 1 < s_end && __builtin_amdgcn_readlane(m_pid, i + 1) == pa.pid0) {
      XPair P;
      issue_pair(s, P);
      sum_pair(s, P);
      s += 2;
      continue;
    }
    const int qb = __builtin_amdgcn_readlane(m_qb, i);
    const int row = s * 64 + lane;
    const bool valid = row < sa.n_rows;
    double acc = (a.init && valid) ? a.init[row] : 0.0;
    double self = 0.0;      // x[row] for the CG dot product: the middle run when that is the diagonal
    bool have_self = false;
    if (pid == pa.pid0) {
      // ---- one run-pattern slice: nine runs of three, all 64 rows exist; 7 quads of codes per lane
      XRun C;
      uchar4 k[7];
      issue(s, C);
      if constexpr (!RC) {
#pragma unroll
        for (int u = 0; u < 7; ++u) k[u] = kbase[(size_t)(qb + u) * 64];
      }
      const double *cw = ctab;  // RC: the row's 27 coefficients
      if constexpr (RC) cw = ctab + C.cls * 27;
      // dictionary values of run u+1 are looked up before the products of run u are summed
      auto code_at = [&](int j) -> int {
        const uchar4 w4 = k[j >> 2];
        const int e = j & 3;
        return e == 0 ? w4.x : e == 1 ? w4.y : e == 2 ? w4.z : w4.w;
      };
      auto coef = [&](int j) -> double {
        if constexpr (RC) return cw[j];
        else return dict[code_at(j)];
      };
      double w0 = coef(0), w1 = coef(1), w2 = coef(2);
      double el = C.edge, er = C.edge;
#pragma unroll
      for (int u = 0; u < 9; ++u) {
        double n0 = 0.0, n1 = 0.0, n2 = 0.0;
        if (u < 8) { n0 = coef(3 * u + 3); n1 = coef(3 * u + 4); n2 = coef(3 * u + 5); }
        const double left = shift_in_from_left(C.m[u], el);
        const double right = shift_in_from_right(C.m[u], er);
        if (u < 8) { el = wave_rol1(el); er = wave_ror1(er); }
        acc += w0 * left;
        acc += w1 * C.m[u];
        acc += w2 * right;
        w0 = n0; w1 = n1; w2 = n2;
      }
      if (pa.centre[4] == 0) { self = C.m[4]; have_self = true; }
    } else {
      // ---- any other slice: quad by quad, the columns and codes of the next quad in flight while
      // the gathers of this one are summed
      const int qe = __builtin_amdgcn_readlane(m_qe, i);
      const int base = pa.col16 ? sa.sbase[s] : 0;
      const int32_t *pat = pid >= 0 ? sa.pat + (size_t)pid * 32 : nullptr;
      auto cols_of = [&](int q, int (&c)[4]) {
        if (pat) {
          const int j = (q - qb) * 4;
          c[0] = row + pat[j]; c[1] = row + pat[j + 1]; c[2] = row + pat[j + 2]; c[3] = row + pat[j + 3];
        } else if (pa.col16) {
          const ushort4 Cq = (reinterpret_cast<const ushort4 *>(sa.cols) + lane)[(size_t)q * 64];
          c[0] = base + Cq.x; c[1] = base + Cq.y; c[2] = base + Cq.z; c[3] = base + Cq.w;
        } else {
          const int4 Cq = (reinterpret_cast<const int4 *>(sa.cols) + lane)[(size_t)q * 64];
          c[0] = Cq.x; c[1] = Cq.y; c[2] = Cq.z; c[3] = Cq.w;
        }
      };
      int c[4] = {0, 0, 0, 0};
      uchar4 kq{0, 0, 0, 0};
      if (qb < qe) { kq = kbase[(size_t)qb * 64]; cols_of(qb, c); }
      for (int q = qb; q < qe; ++q) {
        const double v0 = X((size_t)c[0]), v1 = X((size_t)c[1]), v2 = X((size_t)c[2]), v3 = X((size_t)c[3]);
        const uchar4 k0 = kq;
        if (q + 1 < qe) { kq = kbase[(size_t)(q + 1) * 64]; cols_of(q + 1, c); }
        acc += dict[k0.x] * v0; acc += dict[k0.y] * v1; acc += dict[k0.z] * v2; acc += dict[k0.w] * v3;
      }
    }
    if (valid) finish(row, acc, self, have_self);
    ++s;
  }
  if constexpr (CG != 0) {
    const double sblock = block_sum(dot_acc, red);
    if (threadIdx.x == 0) a.part_out[blockIdx.x] = sblock;
  }
}

// Distributed coarse CG: opens the iteration and forms d = beta d - g on the owned range;
// the ghost entries of d arrive by halo exchange before the SpMV (CG = 2).
struct CGDirArgs {
  double *d;            // out: direction of this iteration
  const double *d_old;  // direction of the previous one (may alias d)
  const double *g;
  int64_t n;
  CGState *st;
  const double *part_in;
  int n_part_in;
  double tol;
  int maxit;
  // peer transport: the halo entries of d go straight into the neighbours' vectors
  PeerCG pc;
  int n_nb;                         // neighbours that receive entries of my d
  int nb_rank[kPeerRanks];
  const int32_t *send_idx;          // owned rows to send, neighbour after neighbour
  int send_off[kPeerRanks + 1];
  double *peer_ghost[kPeerRanks];   // neighbour i: where my segment starts in ITS direction vector of ring slot 0
  long long peer_stride[kPeerRanks];  // doubles between its ring slots
  unsigned int *cnt;                // workgroups that have pushed (the last one publishes the tags)
  int n_push_wg;                    // workgroups [0, n_push_wg) push halo entries besides their share of d
};
__global__ __launch_bounds__(kThreads) void cg_direction_kernel(CGDirArgs a) {
  __shared__ double red[4];
  if (a.st->done) return;
  const int64_t n2 = a.n >> 1;
  const double2 *o2 = reinterpret_cast<const double2 *>(a.d_old);
  const double2 *g2 = reinterpret_cast<const double2 *>(a.g);
  double2 *d2 = reinterpret_cast<double2 *>(a.d);
  // the first pair of every thread is on its way while the partials are reduced and the iteration is opened
  const int64_t i0 = (int64_t)blockIdx.x * kThreads + threadIdx.x, stride = (int64_t)gridDim.x * kThreads;
  double2 ov{0.0, 0.0}, gv{0.0, 0.0};
  if (i0 < n2) { ov = o2[i0]; gv = g2[i0]; }
  double beta = 0.0;
  const int it = a.st->it_k2;
  if (!cg_open_iteration(a.st, a.part_in, a.n_part_in, a.tol, a.maxit, red, &beta)) return;
  for (int64_t i = i0; i < n2; i += stride) {
    if (i != i0) { ov = o2[i]; gv = g2[i]; }
    double2 dv;
    dv.x = beta * ov.x - gv.x; dv.y = beta * ov.y - gv.y;
    d2[i] = dv;
  }
  if ((a.n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = a.n - 1;
    a.d[i] = beta * a.d_old[i] - a.g[i];
  }
  if (a.pc.area && (int)blockIdx.x < a.n_push_wg) {
    // the entries of d my neighbours need, formed again from d_old and g and stored into THEIR vectors
    for (int i = 0; i < a.n_nb; ++i) {
      double *dst = a.peer_ghost[i] + (long long)(it & 7) * a.peer_stride[i];
      const int cnt = a.send_off[i + 1] - a.send_off[i];
      for (int k = blockIdx.x * kThreads + threadIdx.x; k < cnt; k += a.n_push_wg * kThreads) {
        const int li = a.send_idx[a.send_off[i] + k];
        dst[k] = beta * a.d_old[li] - a.g[li];
      }
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(a.cnt, 1u) == (unsigned)a.n_push_wg - 1u) {
      *a.cnt = 0;
      for (int i = 0; i < a.n_nb; ++i)
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(a.pc.peer_area[a.nb_rank[i]] + 2048 + (it & 7) * 64) + a.pc.me,
                           a.pc.tag0 + (unsigned long long)it + 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// One workgroup: this rank's sum of the partials (fixed order) -> every rank's slot (value, then the tag) -> wait for
// everybody's tags -> out[0] = the total, added in rank order.  The consumer kernel reads out[] as a one-element
// "partials" array.  kind 0: |g|^2 that opens iteration it (after the init kernel: it = 0; after an update: it =
// iterations completed); kind 1: d.h of iteration it.
struct PeerSumArgs {
  PeerCG pc;
  const double *part;
  int n_part, kind, from_init;
  const CGState *st;
  double *out;
};
__global__ __launch_bounds__(kThreads) void peer_allsum_kernel(PeerSumArgs a) {
  __shared__ double red[4];
  if (!a.from_init && a.st->done) return;
  if (*(volatile int *)a.pc.abort_flag) return;
  const int it = a.kind == 0 ? (a.from_init ? 0 : a.st->it_k2) : a.st->it_k1;
  const double s = reduce_partials(a.part, a.n_part, red);
  const unsigned long long tag = a.pc.tag0 + (unsigned long long)it + (a.kind == 1 ? 1ull : 0ull);
  const int off = (a.kind * 8 + (it & 7)) * 128;
  if ((int)threadIdx.x < a.pc.n_ranks) reinterpret_cast<double *>(a.pc.peer_area[threadIdx.x] + off)[a.pc.me] = s;
  __threadfence_system();
  __syncthreads();
  if ((int)threadIdx.x < a.pc.n_ranks)
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(a.pc.peer_area[threadIdx.x] + off + 64) + a.pc.me, tag, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  if (!peer_cg_wait(a.pc, peer_cg_tags(a.pc, a.kind, it), (1u << a.pc.n_ranks) - 1u, tag)) return;
  if (threadIdx.x == 0) {
    const double *v = peer_cg_vals(a.pc, a.kind, it);
    double tot = __builtin_nontemporal_load(v);
    for (int r = 1; r < a.pc.n_ranks; ++r) tot += __builtin_nontemporal_load(v + r);
    a.out[0] = tot;
  }
}

// One workgroup, in front of the SpMV of iteration it: the neighbours' direction kernels have stored their entries of d
__global__ __launch_bounds__(64) void peer_wait_halo_kernel(PeerCG pc, const CGState *st) {
  if (st->done || *(volatile int *)pc.abort_flag) return;
  const int it = st->it_k1;
  const unsigned long long *tags = reinterpret_cast<const unsigned long long *>(pc.area + 2048 + (it & 7) * 64);
  (void)peer_cg_wait(pc, tags, pc.nb_mask, pc.tag0 + (unsigned long long)it + 1ull);
}

// ---------------------------------------------------------------- coarse CG: init + update

struct CGInitArgs {
  const double *b;
  double *x, *g, *d0, *d1;
  int64_t n;
  CGState *st;
  double *part_gg;
};

// x = 0 ; g = -b ; d buffers = 0 ; partials of |g|^2 ; reset state          (SolverCG entry, x == 0)
__global__ __launch_bounds__(kThreads) void cg_init_kernel(CGInitArgs a) {
  __shared__ double red[4];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * kThreads) {
    const double gi = -1.0 * a.b[i];
    a.g[i] = gi;
    a.x[i] = 0.0;
    a.d0[i] = 0.0;
    a.d1[i] = 0.0;
    acc += gi * gi;
  }
  const double s = block_sum(acc, red);
  if (threadIdx.x == 0) a.part_gg[blockIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    a.st->gh[0] = a.st->gh[1] = 0.0;
    a.st->res0 = a.st->res = 0.0;
    a.st->it_k1 = a.st->it_k2 = 0;
    a.st->done = 0; a.st->status = 0; a.st->iters = 0;
  }
}

struct CGUpdateArgs {
  double *x, *g;
  const double *d, *h;
  int64_t n;
  CGState *st;
  const double *part_dh;
  int n_part_dh;
  double *part_gg;
};

// alpha = gh / (d.h) ; x += alpha d ; g += alpha h ; partials of |g|^2
__global__ __launch_bounds__(kThreads) void cg_update_kernel(CGUpdateArgs a) {
  __shared__ double red[4];
  if (a.st->done) return;
  const int64_t n2 = a.n >> 1;
  const double2 *d2 = reinterpret_cast<const double2 *>(a.d);
  const double2 *h2 = reinterpret_cast<const double2 *>(a.h);
  double2 *x2 = reinterpret_cast<double2 *>(a.x);
  double2 *g2 = reinterpret_cast<double2 *>(a.g);
  // the first pairs of every thread are on their way while the partials are reduced
  const int64_t i0 = (int64_t)blockIdx.x * kThreads + threadIdx.x, stride = (int64_t)gridDim.x * kThreads;
  double2 dv{0.0, 0.0}, hv{0.0, 0.0}, xv{0.0, 0.0}, gv{0.0, 0.0};
  if (i0 < n2) { dv = d2[i0]; hv = h2[i0]; xv = x2[i0]; gv = g2[i0]; }
  const double dh = reduce_partials(a.part_dh, a.n_part_dh, red);
  const int it = a.st->it_k1;
  const double alpha = a.st->gh[it & 1] / dh;
  double acc = 0.0;
  for (int64_t i = i0; i < n2; i += stride) {
    if (i != i0) { dv = d2[i]; hv = h2[i]; xv = x2[i]; gv = g2[i]; }
    xv.x += alpha * dv.x; xv.y += alpha * dv.y;
    gv.x += alpha * hv.x; gv.y += alpha * hv.y;
    x2[i] = xv; g2[i] = gv;
    acc += gv.x * gv.x;
    acc += gv.y * gv.y;
  }
  if ((a.n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = a.n - 1;
    a.x[i] += alpha * a.d[i];
    const double gi = a.g[i] + alpha * a.h[i];
    a.g[i] = gi;
    acc += gi * gi;
  }
  const double s = block_sum(acc, red);
  if (threadIdx.x == 0) a.part_gg[blockIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) a.st->it_k2 = it + 1;
}

// Three-kernel variant: the update touches g only and leaves alpha in the state; x += alpha d is
// applied for kXRing iterations at once by cg_xflush_kernel from a ring of direction vectors
// (same additions in the same order: x is not read inside the iteration).
struct CGUpdateGArgs {
  double *g;
  const double *h;
  int64_t n;
  CGState *st;
  const double *part_dh;
  int n_part_dh;
  double *part_gg;
};
__global__ __launch_bounds__(kThreads) void cg_update_g_kernel(CGUpdateGArgs a) {
  __shared__ double red[4];
  if (a.st->done) return;
  const int64_t n2 = a.n >> 1;
  const double2 *h2 = reinterpret_cast<const double2 *>(a.h);
  double2 *g2 = reinterpret_cast<double2 *>(a.g);
  // the first pair of every thread (most of the vector at this grid) is on its way while the partials are reduced
  const int64_t i0 = (int64_t)blockIdx.x * kThreads + threadIdx.x, stride = (int64_t)gridDim.x * kThreads;
  double2 hv{0.0, 0.0}, gv{0.0, 0.0};
  if (i0 < n2) { hv = h2[i0]; gv = g2[i0]; }
  const int it = a.st->it_k1;
  const double dh = reduce_partials(a.part_dh, a.n_part_dh, red);
  const double alpha = a.st->gh[it & 1] / dh;
  double acc = 0.0;
  for (int64_t i = i0; i < n2; i += stride) {
    if (i != i0) { hv = h2[i]; gv = g2[i]; }
    gv.x += alpha * hv.x; gv.y += alpha * hv.y;
    g2[i] = gv;
    acc += gv.x * gv.x;
    acc += gv.y * gv.y;
  }
  if ((a.n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = a.n - 1;
    const double gi = a.g[i] + alpha * a.h[i];
    a.g[i] = gi;
    acc += gi * gi;
  }
  const double s = block_sum(acc, red);
  if (threadIdx.x == 0) a.part_gg[blockIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    a.st->alpha[it % kXRing] = alpha;
    a.st->it_k2 = it + 1;
  }
}

// x += sum_{i in [lo, min(completed iterations, upto))} alpha_i d_i, one iteration after the other per element
struct CGXFlushArgs {
  double *x;
  const double *ring[kXRing];  // direction of iteration i lives in ring[i % kXRing]
  int64_t n;
  const CGState *st;
  int lo, upto;
};
__global__ __launch_bounds__(kThreads) void cg_xflush_kernel(CGXFlushArgs a) {
  const int hi = min(a.st->it_k2, a.upto);  // it_k2 does not change while this kernel runs (stream order)
  if (hi <= a.lo) return;
  const int64_t n2 = a.n >> 1;
  double2 *x2 = reinterpret_cast<double2 *>(a.x);
  for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < n2; e += (int64_t)gridDim.x * kThreads) {
    double2 xv = x2[e];
    for (int i = a.lo; i < hi; ++i) {
      const double alpha = a.st->alpha[i % kXRing];
      const double2 dv = reinterpret_cast<const double2 *>(a.ring[i % kXRing])[e];
      xv.x += alpha * dv.x; xv.y += alpha * dv.y;
    }
    x2[e] = xv;
  }
  if ((a.n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t e = a.n - 1;
    double xv = a.x[e];
    for (int i = a.lo; i < hi; ++i) xv += a.st->alpha[i % kXRing] * a.ring[i % kXRing][e];
    a.x[e] = xv;
  }
}

// ---------------------------------------------------------------- BLAS-1 / gather / scatter

__global__ __launch_bounds__(kThreads) void vec_equ_kernel(double *y, double a, const double *x, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) y[i] = a * x[i];
}
__global__ __launch_bounds__(kThreads) void vec_add_kernel(double *y, double a, const double *x, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) y[i] += a * x[i];
}
__global__ __launch_bounds__(kThreads) void vec_sadd_kernel(double *y, double s, double a, const double *x, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads)
    y[i] = s * y[i] + a * x[i];
}
// y = (a x) * z   (Jacobi apply: (omega r) invd)
__global__ __launch_bounds__(kThreads) void vec_scale_mul_kernel(double *y, double a, const double *x, const double *z, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads)
    y[i] = (a * x[i]) * z[i];
}
// Chebyshev start: w = (r invd) / theta ; y = w
__global__ __launch_bounds__(kThreads) void cheb_first_kernel(double *y, double *w, const double *r, const double *invd, double theta, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
    const double wi = (r[i] * invd[i]) / theta;
    w[i] = wi;
    y[i] = wi;
  }
}
// dst[idx_dst[k]] = src[idx_src[k]]   (copy_to_mg / copy_from_mg / halo pack)
__global__ __launch_bounds__(kThreads) void gather_scatter_kernel(double *dst, const int32_t *idx_dst, const double *src,
                                                                 const int32_t *idx_src, int64_t n) {
  for (int64_t k = (int64_t)blockIdx.x * kThreads + threadIdx.x; k < n; k += (int64_t)gridDim.x * kThreads) {
    const int64_t id = idx_dst ? idx_dst[k] : k;
    const int64_t is = idx_src ? idx_src[k] : k;
    dst[id] = src[is];
  }
}

// partial dot: out[blockIdx] ; finalised by reduce_final_kernel
__global__ __launch_bounds__(kThreads) void dot_partial_kernel(const double *x, const double *y, int64_t n, double *part) {
  __shared__ double red[4];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) acc += x[i] * y[i];
  const double s = block_sum(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
// partials of (sum |x|, sum x^2, max |x|, count of nonzeros) -> part[4 * grid]
__global__ __launch_bounds__(kThreads) void norms_partial_kernel(const double *x, int64_t n, double *part) {
  __shared__ double red[4];
  double s1 = 0.0, s2 = 0.0, mx = 0.0, nz = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
    const double v = x[i], av = fabs(v);
    s1 += av; s2 += v * v; mx = fmax(mx, av); nz += (v != 0.0) ? 1.0 : 0.0;
  }
  const double r1 = block_sum(s1, red);
  const double r2 = block_sum(s2, red);
  const double r3 = block_max(mx, red);
  const double r4 = block_sum(nz, red);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = r1;
    part[gridDim.x + blockIdx.x] = r2;
    part[2 * gridDim.x + blockIdx.x] = r3;
    part[3 * gridDim.x + blockIdx.x] = r4;
  }
}
// single workgroup: out[j] = reduce(part[j*n .. j*n+n)) ; op 0 = sum, 1 = max (per j via mask)
__global__ __launch_bounds__(kThreads) void reduce_final_kernel(const double *part, int n, int n_out, unsigned max_mask, double *out) {
  __shared__ double red[4];
  for (int j = 0; j < n_out; ++j) {
    const double *p = part + (int64_t)j * n;
    double r;
    if ((max_mask >> j) & 1u) {
      double m = 0.0;
      for (int i = threadIdx.x; i < n; i += kThreads) m = fmax(m, p[i]);
      r = block_max(m, red);
    } else {
      r = reduce_partials(p, n, red);
    }
    if (threadIdx.x == 0) out[j] = r;
  }
}

// ---------------------------------------------------------------- Gaussian charge density
// rho(x_q) = 4 pi / (r_c^3 pi^1.5) sum_k q_k exp(-|x_q - x_k|^2 / r_c^2) at the quadrature points of
// every active cell (reference: src/step-50.cc:509-575), the sum running over the atoms whose
// distance to ANY vertex of the cell's ROOT cell is below cutoff (the reference's per-cell atom
// lists, :260-306, inherited unchanged by children, :441-450) or over all atoms.  One wavefront
// per cell: lanes stride over the candidate atoms of the neighbouring bins, evaluate the same
// predicate (nearest root-cell vertex, direction by direction), accumulate up to 8 quadrature
// points at a time in registers and combine with a shuffle reduction.
struct DensityArgs {
  const double *cell_lo;    // [n_cells * 3] lower corner of the cell
  const double *cell_h;     // [n_cells]
  const double *root_lo;    // [n_cells * 3] lower corner of the cell's root cell
  double root_h;
  const double *atom_xyz;   // [n_atoms * 3]
  const double *atom_q;
  int n_atoms;
  double bin_lo0, bin_lo1, bin_lo2, bin_size;
  int bin_n0, bin_n1, bin_n2;
  const int32_t *bin_ptr;
  const int32_t *bin_items;
  double cutoff, r_c;
  int use_lists;
  const double *qp;         // [nq * 3] quadrature points on the unit cell
  int nq, n_cells;
  double *dens;             // [n_cells * nq]
};

__global__ __launch_bounds__(kThreads) void charge_density_kernel(DensityArgs a) {
  const int lane = threadIdx.x & 63;
  const int cell = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (cell >= a.n_cells) return;
  const double constant_value = 4.0 * M_PI / (a.r_c * a.r_c * a.r_c * pow(M_PI, 1.5));
  const double inv = 1.0 / (a.r_c * a.r_c);
  double lo[3], rlo[3], rhi[3];
  for (int d = 0; d < 3; ++d) {
    lo[d] = a.cell_lo[3 * (size_t)cell + d];
    rlo[d] = a.root_lo[3 * (size_t)cell + d];
    rhi[d] = rlo[d] + a.root_h;
  }
  const double h = a.cell_h[cell];
  int b0[3] = {0, 0, 0}, b1[3] = {0, 0, 0};
  const double blo[3] = {a.bin_lo0, a.bin_lo1, a.bin_lo2};
  const int bn[3] = {a.bin_n0, a.bin_n1, a.bin_n2};
  if (a.use_lists) {
    for (int d = 0; d < 3; ++d) {
      b0[d] = max((int)floor((rlo[d] - a.cutoff - blo[d]) / a.bin_size), 0);
      b1[d] = min((int)floor((rhi[d] + a.cutoff - blo[d]) / a.bin_size), bn[d] - 1);
    }
  }
  for (int qb = 0; qb < a.nq; qb += 8) {
    const int nqb = min(8, a.nq - qb);
    double xq[8][3], acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      acc[j] = 0.0;
      for (int d = 0; d < 3; ++d) xq[j][d] = j < nqb ? lo[d] + h * a.qp[3 * (qb + j) + d] : 0.0;
    }
    auto add_atom = [&](int k) {
      const double ax = a.atom_xyz[3 * (size_t)k], ay = a.atom_xyz[3 * (size_t)k + 1], az = a.atom_xyz[3 * (size_t)k + 2];
      if (a.use_lists) {
        const double at[3] = {ax, ay, az};
        double d2 = 0.0;
        for (int d = 0; d < 3; ++d) {
          const double dl = at[d] - rlo[d], dh = at[d] - rhi[d];
          const double m = fabs(dl) <= fabs(dh) ? dl : dh;
          d2 += m * m;
        }
        if (!(sqrt(d2) < a.cutoff)) return;
      }
      const double qk = a.atom_q[k];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j < nqb) {
          const double dx = ax - xq[j][0], dy = ay - xq[j][1], dz = az - xq[j][2];
          const double r = sqrt(dx * dx + dy * dy + dz * dz);  // the reference squares the distance again
          acc[j] += constant_value * exp(-(r * r) * inv) * qk;
        }
      }
    };
    if (a.use_lists) {
      for (int z = b0[2]; z <= b1[2]; ++z)
        for (int y = b0[1]; y <= b1[1]; ++y)
          for (int x = b0[0]; x <= b1[0]; ++x) {
            const int64_t b = (int64_t)x + bn[0] * ((int64_t)y + (int64_t)bn[1] * z);
            const int ks = a.bin_ptr[b], ke = a.bin_ptr[b + 1];
            for (int k = ks + lane; k < ke; k += 64) add_atom(a.bin_items[k]);
          }
    } else {
      for (int k = lane; k < a.n_atoms; k += 64) add_atom(k);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const double s = wave_sum(acc[j]);
      if (lane == 0 && j < nqb) a.dens[(size_t)cell * a.nq + qb + j] = s;
    }
  }
}


// ---------------------------------------------------------------- right-hand side from device-resident densities
// assemble_system's F (src/step-50.cc:813-828): per cell F_i = sum_q phi_i(x_q) rho(x_q) w_q JxW in the reference's
// operation order, then the Dirichlet terms (F_slot -= K_ij g_j, one list entry per subtraction, applied in list order by
// the thread that owns the slot's run), then per DoF the sum over its (cell, vertex) slots in cell order with the
// constraint weights.  Everything sequential per output value: deterministic, the host loop's bits.
struct RhsArgs {
  const double *dens;  // [cells][nq]
  int64_t n_cells;
  int nq, nv;
  const uint8_t *cell_level;
  double shape[512 * 8];  // [q][i] (8 slots per q; 2D uses the first four); up to 8^3 quadrature points
  double weight[512];
  double jxw[16];        // by level
  double *F;             // [cells][nv]
  int64_t n_terms;
  const int32_t *term_slot;  // ascending
  const double *term_value;
  int64_t n_dofs;
  const int32_t *dof_ptr, *entry_slot;
  const uint8_t *entry_coef;  // 0: the slot's F as it is; c > 0: coef[c] * F
  double coef[256];
  double *rhs;
};
__global__ __launch_bounds__(kThreads) void rhs_cell_kernel(const RhsArgs *ap) {
  const RhsArgs &a = *ap;
  for (int64_t c = (int64_t)blockIdx.x * kThreads + threadIdx.x; c < a.n_cells; c += (int64_t)gridDim.x * kThreads) {
    const double jxw = a.jxw[a.cell_level[c] & 15];
    double F[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int q = 0; q < a.nq; ++q) {
      const double d = a.dens[c * a.nq + q], w = a.weight[q];
      for (int i = 0; i < a.nv; ++i) F[i] += a.shape[q * 8 + i] * d * w * jxw;
    }
    for (int i = 0; i < a.nv; ++i) a.F[c * a.nv + i] = F[i];
  }
}
// the subtractions of one slot form a run in the list: the thread at the run's first entry applies all of them, in order
__global__ __launch_bounds__(kThreads) void rhs_terms_kernel(const RhsArgs *ap) {
  const RhsArgs &a = *ap;
  for (int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x; t < a.n_terms; t += (int64_t)gridDim.x * kThreads) {
    const int32_t slot = a.term_slot[t];
    if (t > 0 && a.term_slot[t - 1] == slot) continue;
    double f = a.F[slot];
    for (int64_t u = t; u < a.n_terms && a.term_slot[u] == slot; ++u) f -= a.term_value[u];
    a.F[slot] = f;
  }
}
__global__ __launch_bounds__(kThreads) void rhs_gather_kernel(const RhsArgs *ap) {
  const RhsArgs &a = *ap;
  for (int64_t d = (int64_t)blockIdx.x * kThreads + threadIdx.x; d < a.n_dofs; d += (int64_t)gridDim.x * kThreads) {
    double acc = 0.0;
    for (int32_t e = a.dof_ptr[d]; e < a.dof_ptr[d + 1]; ++e) {
      const double f = a.F[a.entry_slot[e]];
      const int c = a.entry_coef[e];
      acc += c == 0 ? f : a.coef[c] * f;
    }
    a.rhs[d] = acc;
  }
}

// ---------------------------------------------------------------- HBM calibration (measurement only)
// Pure streaming read (16 B / lane, grid-stride) and copy: the ceiling the SpMV is compared with
// on the device it actually runs on (bench.py reports it next to the 8 TB/s spec figure).
__global__ __launch_bounds__(kThreads) void stream_read_kernel(const double2 *x, int64_t n2, double *part) {
  __shared__ double red[4];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kThreads) {
    const double2 v = x[i];
    acc += v.x + v.y;
  }
  const double s = block_sum(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(kThreads) void stream_copy_kernel(const double2 *x, double2 *y, int64_t n2) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kThreads) y[i] = x[i];
}

// ---------------------------------------------------------------- SSOR (level-scheduled SGS), generic fallback
// (the production path is the wavefront sweep of gmg_sgs.hpp; this one takes any row width straight from the CSR copy)
// Ifpack's symmetric Gauss-Seidel is sequential in the local row order; rows without mutual
// coupling form "stages" (computed on the host), so sweeping stage by stage reproduces the
// sequential result exactly.  One workgroup (1024 threads = 32 half-waves) owns one block of
// consecutive rows: with a single block this is the exact 1-rank smoother; with B blocks the
// couplings to other blocks are dropped, which is precisely what the reference's smoother does
// on B MPI ranks (block Jacobi of rank-local SGS).  A half-wave handles a row:
// 32 lanes fetch val, col and y[col] in parallel (one dependent round trip instead of 27),
// then the products are added IN CSR ORDER through shuffles -> bit-identical to the sequential sum.
struct SgsArgs {
  const int32_t *rowptr;
  const int32_t *col;
  const double *val;
  const double *invd;
  const int32_t *block_row;    // n_blocks + 1: row range of each block
  const int32_t *block_stage;  // n_blocks + 1: offsets into stage_ptr (stages of each block)
  const int32_t *stage_ptr;    // per block: (n_stages_b + 1) offsets into stage_rows, concatenated
  const int32_t *stage_rows;   // rows ordered by (block, stage)
  double omega;
  const double *r;
  double *y;
};

template <bool BACKWARD>
__global__ __launch_bounds__(1024) void sgs_sweep_kernel(SgsArgs a) {
  const int b = blockIdx.x;
  const int rb = a.block_row[b], re = a.block_row[b + 1];
  const int s0 = a.block_stage[b] + b, s1 = a.block_stage[b + 1] + b;  // this block's slice of stage_ptr
  const int n_stages = s1 - s0;
  const int hw = threadIdx.x >> 5, hl = threadIdx.x & 31;  // half-wave id, lane in it
  for (int st = 0; st < n_stages; ++st) {
    const int stage = BACKWARD ? n_stages - 1 - st : st;
    const int qb = a.stage_ptr[s0 + stage], qe = a.stage_ptr[s0 + stage + 1];
    for (int q = qb + hw; q < qe; q += 32) {
      const int i = a.stage_rows[q];
      const int k0 = a.rowptr[i], k1 = a.rowptr[i + 1];
      double acc = 0.0;
      for (int kb = k0; kb < k1; kb += 32) {
        double prod = 0.0;
        const int k = kb + hl;
        if (k < k1) {
          const int c = a.col[k];
          if (c >= rb && c < re) prod = a.val[k] * a.y[c];  // other blocks' columns are dropped
        }
#pragma unroll
        for (int t = 0; t < 32; ++t) acc += __shfl(prod, t, 32);  // in CSR order; padding adds +0.0
      }
      if (hl == 0) a.y[i] += a.omega * (a.r[i] - acc) * a.invd[i];
    }
    __threadfence_block();
    __syncthreads();
  }
}

}  // namespace gmg
