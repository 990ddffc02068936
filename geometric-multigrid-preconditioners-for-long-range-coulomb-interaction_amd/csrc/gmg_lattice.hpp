// gmg_lattice.hpp -- the level-0 operator (a 27-point stencil on an nx x ny x nz vertex lattice, lexicographic DoFs)
// applied plane by plane with a sliding window of x lines in registers.
//
// Reference: LaplaceOperator / mg_matrices[0].vmult inside the coarse CG (src/step-50.cc:962-967) and the cycle-0 system
// matrix (:991); SURVEY.md 8(a) A3, 8(f) N4 (row classes instead of stored entries).  Same products, same order as the CSR
// row (ascending columns = dz, dy, dx lexicographic), absent neighbours contribute an explicit +0.0 * x: bit-identical to
// oracle/gmg_oracle.c:csr_spmv for finite x.
//
// Why another kernel.  spmv_sellp_kernel (gmg_device.hpp) reads, for every unit of 128 rows, all nine x lines of the
// stencil (9 + 1 + 1 vector-memory instructions, 36 lane shifts) and was bound by instruction issue, not by bytes
// (profiles/r02_pmc_spmv_sellp.txt: 147 VALU and 6.5 VMEM instructions per 64 rows; 0.24-0.34 of the HBM peak).  On a
// lattice the nine lines of row block b + nxy (one plane up) are six of the lines of block b plus three new ones.  So a
// wave here owns a COLUMN of units  b_k = R0 + c * 124 + k * nxy,  k = k0 .. k1,  and marches along z:
//   * window: 3 planes x 3 lines, each line = the lane's pair {x[r], x[r + 1]} plus the two lane-shifted copies
//     (x[r - 1] for the even row, x[r + 2] for the odd row); per step only the plane dz = +1 is new: 3 x 16-byte loads
//     per lane and 12 DPP moves instead of 9 loads and 36 moves; a fourth plane is in flight while the step is summed;
//   * lane l holds rows r = b - 2 + 2 l, r + 1; lanes 1..62 produce 124 rows, lanes 0 and 63 only feed the shifts:
//     no edge loads, no lane rotations, no scalar traffic (3 % of the lanes' work is spent on the overlap);
//   * coefficients: one class byte per row, 27 doubles per class from an LDS table (as in the RC variant before);
//   * XCD-aware: blockIdx % 8 owns a contiguous z-slab (its planes stay in that XCD's L2: three planes of 201^2 are 1 MB),
//     consecutive waves of a workgroup take neighbouring columns of the same planes;
//   * the interior is what the numbering leaves regular: with a lexicographic numbering every row whose 27 neighbours
//     exist; with deal.II's cell-by-cell numbering (the host side's: a cell's new vertices in first-touch order) the
//     first three lines of every plane and the first three planes are irregular, so the interior is a WINDOW of W rows
//     that repeats with the plane stride.  Rows outside it -- and with a row-partitioned level 0 the rows that touch
//     ghost columns -- are SELL slices served by the last workgroups of the same launch from the streams of
//     gmg_device.hpp (one launch, one set of reduction partials); interior rows inside such a slice are masked there.
#pragma once
#include "gmg_device.hpp"

namespace gmg {

constexpr int kLatMaxClasses = 128;  // 27 doubles each in LDS: 27.6 KB (125 = the position types of gmg_set_level_matrix_lattice)
constexpr int kLatRowsPerUnit = 124;
constexpr int kLatAhead = 1;         // plane steps of loads in flight beyond the one being summed
constexpr int kLatWavesPerSimd = 4;  // register budget of the kernel (128 VGPRs): the window of planes lives in registers

struct LatArgs {
  SellPatArgs pa;         // SELL streams + vectors (pa.sa.a): the slices outside [R0, R1) are served from them
  const uint8_t *rowcls;  // [n_rows]: class of every row of [R0, R1)
  const double *ctab;     // [n_classes][27]: coefficients in entry order (dz, dy, dx ascending), +0.0 where the row stores nothing
  int n_classes;
  int nx, nxy;            // line and plane stride
  int W;                  // the interior is a window of W rows repeating with the plane stride: rows R0 + k nxy + [0, W), below R1
                          // (W == nxy: one contiguous interior [R0, R1), the last plane step cut by R1)
  int R0, R1;
  int C;                  // columns (units of 124 rows) per plane step = ceil(W / 124)
  int K;                  // plane steps
  int S;                  // segments per XCD slab and column
  int fast_blocks;        // workgroups [0, fast_blocks) march; the others serve gen_slices.  Their waves take the S * C (segment,
                          // column) pairs of their XCD in turn: one each on every lattice up to ~360^3, several beyond (the grid is
                          // capped by the reduction partials)
  const int32_t *gen_slices;
  int n_gen;
  // edge_mode = 1 (operators made by gmg_set_level_matrix_lattice: no SELL streams exist): the rows outside [R0, R1) are
  // served from the class table too, row by row with guarded gathers; n_gen then counts chunks of 64 such rows
  int edge_mode;
  int n_rows;
};

namespace lat {

// lane L <- lane L - 1 (lane 0 keeps `v`: its result is never used) / lane L <- lane L + 1
__device__ __forceinline__ double from_left(double v) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_right(double v) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

struct Plane {       // three lines (dy = -1, 0, +1) of one plane, as one lane sees them
  double2 p[3];      // {x[r + off], x[r + 1 + off]}
  double lf[3];      // x[r - 1 + off]   (left neighbour of the even row)
  double rt[3];      // x[r + 2 + off]   (right neighbour of the odd row)
};

}  // namespace lat

// One SELL slice outside the lattice interior, lane = row, from the streams of gmg_device.hpp (values as 8-bit codes into
// the dictionary, columns from the slice's pattern or from the column stream).  Returns through `finish`.
template <int CG>
__device__ __forceinline__ void lattice_generic_slice(const SellPatArgs &pa, int R0, int R1, int nxy, int W, int s, int lane, const double *dict, double &dot_acc) {
  const SellArgs &sa = pa.sa;
  const SpmvArgs &a = sa.a;
  const int qb = __builtin_amdgcn_readfirstlane(sa.qptr[s]), qe = __builtin_amdgcn_readfirstlane(sa.qptr[s + 1]);
  const int pid = sa.spat ? __builtin_amdgcn_readfirstlane(sa.spat[s]) : -1;
  const int row = s * 64 + lane;
  // rows of the lattice interior inside a slice that also holds other rows belong to the marching waves
  const bool valid = row < sa.n_rows && !(row >= R0 && row < R1 && (row - R0) % nxy < W);
  const uchar4 *kbase = reinterpret_cast<const uchar4 *>(sa.vals) + lane;
  const int base = pa.col16 ? __builtin_amdgcn_readfirstlane(sa.sbase[s]) : 0;
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
  double acc = 0.0;
  int c[4] = {0, 0, 0, 0};
  uchar4 kq{0, 0, 0, 0};
  if (qb < qe) { kq = kbase[(size_t)qb * 64]; cols_of(qb, c); }
  for (int q = qb; q < qe; ++q) {
    const double v0 = a.x[c[0]], v1 = a.x[c[1]], v2 = a.x[c[2]], v3 = a.x[c[3]];
    const uchar4 k0 = kq;
    if (q + 1 < qe) { kq = kbase[(size_t)(q + 1) * 64]; cols_of(q + 1, c); }
    acc += dict[k0.x] * v0; acc += dict[k0.y] * v1; acc += dict[k0.z] * v2; acc += dict[k0.w] * v3;
  }
  if (valid) {
    a.y[row] = acc;
    if constexpr (CG == 2) dot_acc += a.x[row] * acc;
  }
}

// A chunk of 64 rows outside the lattice interior of a pure lattice operator (edge_mode): the first R0 and the last
// n - R1 rows of the numbering.  Lane = row; the 27 coefficients of the row's class against x at the 27 lattice offsets, in
// entry order; an offset that leaves [0, n) reads nothing (its coefficient is +0.0: the neighbour does not exist).
template <int CG>
__device__ __forceinline__ void lattice_edge_chunk(const LatArgs &A, int chunk, int lane, const double *ctab, double &dot_acc) {
  const SpmvArgs &a = A.pa.sa.a;
  const int head = (A.R0 + 63) >> 6;  // chunks of the head rows [0, R0)
  const int row = chunk < head ? chunk * 64 + lane : A.R1 + (chunk - head) * 64 + lane;
  const bool valid = chunk < head ? row < A.R0 : row < A.n_rows;
  if (!valid) return;
  const double *cw = ctab + (int)A.rowcls[row] * 27;
  double acc = 0.0;
#pragma unroll
  for (int j = 0; j < 27; ++j) {
    const int col = row + (j / 9 - 1) * A.nxy + ((j / 3) % 3 - 1) * A.nx + (j % 3 - 1);
    const double xv = (col >= 0 && col < A.n_rows) ? a.x[col] : 0.0;
    acc += cw[j] * xv;
  }
  a.y[row] = acc;
  if constexpr (CG == 2) dot_acc += a.x[row] * acc;
}

// gmg_set_level_matrix_lattice: class byte of every vertex of an nx x ny x nz lattice (position type per direction: on the
// low face, next to it, inside, next to the high face, on it -> 5 x 5 x 5 classes) and 1 / a_ii from the class table.
struct LatticeClassMap { uint8_t cls[125]; };  // position type (tx + 5 ty + 25 tz) -> class (types with equal coefficients share one)
__global__ __launch_bounds__(kThreads) void lattice_rowclass_kernel(uint8_t *rowcls, double *invd, const double *ctab, LatticeClassMap cmap, int nx, int ny, int nz) {
  const long long n = (long long)nx * ny * nz;
  for (long long r = (long long)blockIdx.x * kThreads + threadIdx.x; r < n; r += (long long)gridDim.x * kThreads) {
    const int x = (int)(r % nx), y = (int)((r / nx) % ny), z = (int)(r / ((long long)nx * ny));
    auto type = [](int c, int m) { return c == 0 ? 0 : c == m - 1 ? 4 : c == 1 ? 1 : c == m - 2 ? 3 : 2; };
    const int cls = cmap.cls[type(x, nx) + 5 * type(y, ny) + 25 * type(z, nz)];
    rowcls[r] = (uint8_t)cls;
    invd[r] = 1.0 / ctab[cls * 27 + 13];
  }
}

// CG = 0: y = A x.   CG = 2: the coarse CG's h = A d with the partials of d.h (returns at once when the solve is over).
// MULTI: the marching waves take several (segment, column) pairs each (lattices above ~360^3: the grid is capped by the
// reduction partials); a variant of its own because the loop costs the single-pass form two registers it does not have
template <int CG, bool MULTI = false>
__global__ __launch_bounds__(kThreads, kLatWavesPerSimd) void spmv_lattice_kernel(LatArgs A) {
  __shared__ double red[4];
  __shared__ double dict[256];
  __shared__ double ctab[kLatMaxClasses * 27];
  const SpmvArgs &a = A.pa.sa.a;
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double dot_acc = 0.0;
  if ((int)blockIdx.x < A.fast_blocks) {
    // ---------------------------------------------------------------- marching waves
    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3;
    const int n_wv = A.S * A.C, wv_stride = (A.fast_blocks >> 3) * 4;
    // wave within its XCD: (segment, column), neighbouring columns in one workgroup; a second pass only on lattices whose
    // columns outnumber the grid
    for (int wv = lb * 4 + wid, pass = 0; pass == 0 || (MULTI && wv < n_wv); wv += wv_stride, ++pass) {
    const int s = wv / A.C, c = wv - s * A.C;
    const int kx0 = (A.K * xcd) >> 3, kx1 = (A.K * (xcd + 1)) >> 3;
    int k0 = kx0 + ((kx1 - kx0) * s) / A.S, k1 = kx0 + ((kx1 - kx0) * (s + 1)) / A.S;
    const int rb0 = A.R0 - 2 + c * kLatRowsPerUnit;  // row of lane 0's even slot in plane step 0
    // steps whose unit holds no row below R1 are dropped (only the last plane step can be partial)
    while (k1 > k0 && rb0 + 2 + (k1 - 1) * A.nxy >= A.R1) --k1;
    const bool live = s < A.S && k0 < k1;
    // rows of this column inside the window: offsets [c * 124, min(c * 124 + 124, W))
    const int in_plane = min(kLatRowsPerUnit, A.W - c * kLatRowsPerUnit);  // >= 1 (C = ceil(W / 124))
    const char *xbytes = reinterpret_cast<const char *>(a.x);
    // per step: lane l reads the pair at row rb + 2 min(l, lim), lim = last pair that still starts at a valid row + 1
    auto load_plane = [&](int rb, int dzoff, uint32_t voff, double2 (&dst)[3]) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const long long off = ((long long)rb + dzoff + (j - 1) * A.nx) * 8;  // wave-uniform
        dst[j] = *reinterpret_cast<const double2 *>(xbytes + off + voff);
      }
    };
    // 3 + AH plane slots used in rotation (no register moves): at a step of phase PH the planes dz = -1, 0, +1 sit in slots
    // PH, PH + 1, PH + 2 (mod NS); the planes dz = +1 of the next AH steps are in flight into the slots behind them.  The
    // plane of the NEXT step is waited for, and its lane-shifted copies formed, at the end of this step.
    // AH = 1 (4 slots, 118 VGPRs, 4 waves / SIMD) measured faster than AH = 2 (5 slots, 146 VGPRs, 3 waves / SIMD) at
    // 121^3 (11.8 vs 13.7 us) and at 201^3 (42.4 vs 46.5 us): the occupancy is worth more than the second step in flight.
    constexpr int AH = kLatAhead, NS = 3 + AH, NC = 1 + AH;
    lat::Plane Q[NS];
    int cls[NC] = {};          // class bytes of this step and the AH after it (rotating with the phase: cls[PH % NC])
    uint32_t voff[NC] = {};
    int nlim[NC] = {};
    auto step_limits = [&](int k, uint32_t &vo, int &nl) {
      const int rb = rb0 + k * A.nxy;
      // last valid row of the step relative to rb (rows rb + 2 .. rb + 125 belong to lanes 1..62): inside the column's
      // share of the window and below R1.  A lane beyond it re-reads the pair of the last lane whose rows (or whose left
      // neighbour's right-hand value) are needed: every access stays inside the vectors (allocated with two spare entries)
      nl = min(in_plane + 1, A.R1 - 1 - rb);
      vo = (uint32_t)min(lane, (nl + 1) >> 1) * 16u;
    };
    auto load_cls = [&](int k, uint32_t vo) -> int { return *reinterpret_cast<const unsigned short *>(A.rowcls + rb0 + k * A.nxy + (vo >> 3)); };
    if (live) {
      // prologue: planes dz = -1, 0, +1 of step k0 and the planes dz = +1 of the steps up to k0 + AH - 1, class bytes
      const int rb = rb0 + k0 * A.nxy;
      step_limits(k0, voff[0], nlim[0]);
      load_plane(rb, -A.nxy, voff[0], Q[0].p);
      load_plane(rb, 0, voff[0], Q[1].p);
      load_plane(rb, A.nxy, voff[0], Q[2].p);
      cls[0] = load_cls(k0, voff[0]);
#pragma unroll
      for (int d = 1; d < AH; ++d)
        if (k0 + d < k1) {
          step_limits(k0 + d, voff[d], nlim[d]);
          load_plane(rb + d * A.nxy, A.nxy, voff[d], Q[2 + d].p);
          cls[d] = load_cls(k0 + d, voff[d]);
        }
    }
    if (pass == 0) {
      if constexpr (CG == 2) {
        if (a.st->done) return;  // (read after the prologue loads were issued: one round trip for all of them)
      }
      for (int i = threadIdx.x; i < A.n_classes * 27; i += kThreads) ctab[i] = A.ctab[i];
      __syncthreads();
    }
    if (live) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          Q[q].lf[j] = lat::from_left(Q[q].p[j].y);
          Q[q].rt[j] = lat::from_right(Q[q].p[j].x);
        }
      typedef const __attribute__((address_space(3))) double lds_cdouble;
      const uint32_t tab = (uint32_t)(uintptr_t)(lds_cdouble *)ctab;
      auto coef = [](uint32_t base, int j) -> double { return *reinterpret_cast<lds_cdouble *>((uintptr_t)(base + 8u * (uint32_t)j)); };
      int k = k0;
      auto step = [&](auto phase) {
        constexpr int PH = decltype(phase)::value;      // k - k0 modulo NP: plane slots rotate mod NS, the scalars mod NC
        constexpr int PS = PH % NS, PC = AH == 1 ? 0 : PH % NC;  // (AH = 1: the two sets of scalars are copied, not rotated: four phases of code instead of eight)
        lat::Plane &N1 = Q[(PS + 3) % NS];              // the next step's new plane: arrives during this step (AH = 1: issued now)
        lat::Plane &NA = Q[(PS + 2 + AH) % NS];         // issued now: the new plane of step k + AH
        constexpr int CA = (PC + AH) % NC;
        const int rb = rb0 + k * A.nxy;
        if (k + AH < k1) {
          step_limits(k + AH, voff[CA], nlim[CA]);
          load_plane(rb + AH * A.nxy, A.nxy, voff[CA], NA.p);
          cls[CA] = load_cls(k + AH, voff[CA]);
        } else if constexpr (AH > 1) {
          // (a value that is only conditionally overwritten stays live across the loop in the compiler's eyes -- 36
          // registers' worth with five slots: so the other path overwrites it too)
#pragma unroll
          for (int j = 0; j < 3; ++j) NA.p[j] = double2{0.0, 0.0};
          cls[CA] = 0; voff[CA] = 0; nlim[CA] = 0;
        }
        // ---- sums of the step: rows rb + 2 l (even) and rb + 2 l + 1 (odd), nine runs of three in CSR order
        const int cls_cur = cls[PC];
        uint32_t b0 = tab + (uint32_t)(cls_cur & 0xff) * 216u, b1 = tab + (uint32_t)((cls_cur >> 8) & 0xff) * 216u;
        double c0[3] = {coef(b0, 0), coef(b0, 1), coef(b0, 2)}, c1[3] = {coef(b1, 0), coef(b1, 1), coef(b1, 2)};
        double acc0 = 0.0, acc1 = 0.0, t0 = 0.0, t1 = 0.0;
#pragma unroll
        for (int u = 0; u < 9; ++u) {
          const lat::Plane &W = Q[(PS + u / 3) % NS];
          const int j = u % 3;
          double n0[3] = {0.0, 0.0, 0.0}, n1[3] = {0.0, 0.0, 0.0};
          if (u < 8) {
            // (the coefficient reads of run u + 1 are tied to the sums as they stand when run u starts: left alone the
            // compiler reads all 54 first and spills)
            asm volatile("" : "+v"(b0), "+v"(b1) : "v"(t0), "v"(t1));
            t0 = acc0; t1 = acc1;
#pragma unroll
            for (int e = 0; e < 3; ++e) { n0[e] = coef(b0, 3 * u + 3 + e); n1[e] = coef(b1, 3 * u + 3 + e); }
          }
          acc0 += c0[0] * W.lf[j];
          acc0 += c0[1] * W.p[j].x;
          acc0 += c0[2] * W.p[j].y;
          acc1 += c1[0] * W.p[j].x;
          acc1 += c1[1] * W.p[j].y;
          acc1 += c1[2] * W.rt[j];
#pragma unroll
          for (int e = 0; e < 3; ++e) { c0[e] = n0[e]; c1[e] = n1[e]; }
        }
        // ---- results: lanes 1..62, rows up to the step's last valid one
        const int l2 = 2 * lane;
        const bool inner = lane >= 1 && lane <= 62;
        const bool v0 = inner && l2 <= nlim[PC], v1 = inner && l2 + 1 <= nlim[PC];
        double *yr = a.y + rb + l2;
        if (v1) {
          *reinterpret_cast<double2 *>(yr) = double2{acc0, acc1};
        } else if (v0) {
          yr[0] = acc0;
        }
        if constexpr (CG == 2) {
          // d.h over the rows this lane owns (x is the direction d: the middle line of the middle plane)
          const lat::Plane &M = Q[(PS + 1) % NS];
          dot_acc += v0 ? M.p[1].x * acc0 : 0.0;
          dot_acc += v1 ? M.p[1].y * acc1 : 0.0;
        }
        // the next step's new plane: its lane-shifted copies (this is where its loads are waited for)
        if (AH > 1 || k + 1 < k1) {
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            N1.lf[j] = lat::from_left(N1.p[j].y);
            N1.rt[j] = lat::from_right(N1.p[j].x);
          }
        }
        if constexpr (AH == 1) { cls[0] = cls[1]; voff[0] = voff[1]; nlim[0] = nlim[1]; }
        ++k;
      };
      constexpr int NP = AH == 1 ? NS : NS * NC;  // phases until slots (and scalars) are back where they started
#define LAT_STEP(i) if constexpr (i < NP) { step(std::integral_constant<int, i>{}); if (k >= k1) break; }
      for (;;) {
        LAT_STEP(0) LAT_STEP(1) LAT_STEP(2) LAT_STEP(3) LAT_STEP(4) LAT_STEP(5) LAT_STEP(6) LAT_STEP(7)
        LAT_STEP(8) LAT_STEP(9) LAT_STEP(10) LAT_STEP(11) LAT_STEP(12) LAT_STEP(13) LAT_STEP(14)
      }
#undef LAT_STEP
    }
    }  // (passes)
  } else {
    // ---------------------------------------------------------------- the slices outside the lattice interior
    if constexpr (CG == 2) {
      if (a.st->done) return;
    }
    const int gw = ((int)blockIdx.x - A.fast_blocks) * 4 + wid, n_gw = ((int)gridDim.x - A.fast_blocks) * 4;
    if (A.edge_mode) {
      for (int i = threadIdx.x; i < A.n_classes * 27; i += kThreads) ctab[i] = A.ctab[i];
      __syncthreads();
      for (int i = gw; i < A.n_gen; i += n_gw) lattice_edge_chunk<CG>(A, i, lane, ctab, dot_acc);
    } else {
      dict[threadIdx.x] = A.pa.sa.dict[threadIdx.x];
      __syncthreads();
    }
    for (int i = gw; i < (A.edge_mode ? 0 : A.n_gen); i += n_gw) lattice_generic_slice<CG>(A.pa, A.R0, A.R1, A.nxy, A.W, __builtin_amdgcn_readfirstlane(A.gen_slices[i]), lane, dict, dot_acc);
  }
  if constexpr (CG != 0) {
    const double sblock = block_sum(dot_acc, red);
    if (threadIdx.x == 0) a.part_out[blockIdx.x] = sblock;
  }
}

}  // namespace gmg
