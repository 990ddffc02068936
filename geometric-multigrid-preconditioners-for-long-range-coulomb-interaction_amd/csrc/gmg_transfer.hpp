// gmg_transfer.hpp -- MGTransferPrebuilt::build_matrices on the device (second half).
//
// Reference: mg_transfer.build_matrices(mg_dof_handler), /root/reference/src/step-50.cc:957-958 (inside the "Solve" timer
// opened at :941): for Q1 elements the prolongation P_l (level l -> l + 1) is the trilinear embedding of every refined cell
// into its children, with the columns of coarse boundary DoFs dropped (MGConstrainedDoFs, :704-706); restrict_and_add uses
// its transpose.  The host side builds it cell by cell (csrc/host/laplace_problem.cc: build_transfer); here it is built from
// what the two levels ARE -- the vertex of every level DoF -- because the embedding is purely geometric:
//   a fine vertex f has, per direction, either one coarse parent (f_d a multiple of the coarse spacing: weight 1) or two
//   (f_d half-way between two coarse vertices: weight 1/2 each); its row of P is the tensor product of those, minus the
//   coarse DoFs on the boundary, in ascending column order.  Column c of P (row c of P^T) lists the fine vertices
//   c + {-1, 0, 1}^dim * (spacing / 2) that exist on the fine level, in ascending fine DoF order (the order in which the
//   reference's sequential Tvmult adds them).
// Vertex -> DoF look-ups go through an open-addressing hash table built by a kernel (64-bit keys, atomicCAS); row pointers
// by a one-workgroup scan; rows are sorted in registers (<= 8 / <= 27 entries).  Values are products of 1 and 1/2: exact.
#pragma once
#include "gmg_device.hpp"

namespace gmg {

constexpr unsigned long long kTrEmpty = ~0ull;

__device__ __forceinline__ unsigned long long tr_hash(unsigned long long k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
  return k;
}

// table[mask + 1] keys (kTrEmpty = free), dof[mask + 1]
__global__ __launch_bounds__(kThreads) void tr_table_build_kernel(const unsigned long long *vertex, int64_t n, unsigned long long *keys, int32_t *dof, unsigned long long mask) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
    const unsigned long long k = vertex[i];
    unsigned long long h = tr_hash(k) & mask;
    for (;;) {
      const unsigned long long prev = atomicCAS(&keys[h], kTrEmpty, k);
      if (prev == kTrEmpty || prev == k) { dof[h] = (int32_t)i; break; }  // (a vertex is one DoF of its level: no duplicates)
      h = (h + 1) & mask;
    }
  }
}
__device__ __forceinline__ int32_t tr_lookup(const unsigned long long *keys, const int32_t *dof, unsigned long long mask, unsigned long long k) {
  unsigned long long h = tr_hash(k) & mask;
  for (;;) {
    const unsigned long long q = keys[h];
    if (q == k) return dof[h];
    if (q == kTrEmpty) return -1;
    h = (h + 1) & mask;
  }
}

struct TransferArgs {
  const unsigned long long *fine_vertex, *coarse_vertex;  // x | y << 21 | z << 42 in units of the finest addressable lattice
  const uint8_t *coarse_boundary;
  int64_t n_fine, n_coarse;
  const unsigned long long *fkeys, *ckeys;  // hash tables of the two levels
  const int32_t *fdof, *cdof;
  unsigned long long fmask, cmask;
  int dim;
  unsigned long long half;  // half the coarse spacing = the fine spacing, in lattice units
  int32_t *rowptr;          // counts in, exclusive scan out (n + 1)
  int32_t *col;
  double *val;
};

__device__ __forceinline__ void tr_unpack(unsigned long long k, unsigned long long v[3]) {
  v[0] = k & 0x1FFFFF; v[1] = (k >> 21) & 0x1FFFFF; v[2] = (k >> 42) & 0x1FFFFF;
}
__device__ __forceinline__ unsigned long long tr_pack(const unsigned long long v[3]) { return v[0] | (v[1] << 21) | (v[2] << 42); }

// Row f of P: FILL = false counts, FILL = true writes (ascending columns).
template <bool FILL>
__global__ __launch_bounds__(kThreads) void tr_prolongation_kernel(TransferArgs a) {
  for (int64_t f = (int64_t)blockIdx.x * kThreads + threadIdx.x; f < a.n_fine; f += (int64_t)gridDim.x * kThreads) {
    unsigned long long v[3];
    tr_unpack(a.fine_vertex[f], v);
    unsigned long long par[3][2];
    int np[3] = {1, 1, 1};
    for (int d = 0; d < 3; ++d) {
      par[d][0] = v[d]; par[d][1] = v[d];
      if (d < a.dim && (v[d] & (2 * a.half - 1)) != 0) { par[d][0] = v[d] - a.half; par[d][1] = v[d] + a.half; np[d] = 2; }
    }
    int32_t c[8];
    double w[8];
    int cnt = 0;
    for (int iz = 0; iz < np[2]; ++iz)
      for (int iy = 0; iy < np[1]; ++iy)
        for (int ix = 0; ix < np[0]; ++ix) {
          const unsigned long long p[3] = {par[0][ix], par[1][iy], par[2][iz]};
          const int32_t cd = tr_lookup(a.ckeys, a.cdof, a.cmask, tr_pack(p));
          if (cd < 0 || a.coarse_boundary[cd]) continue;
          double wt = 1.0;
          for (int d = 0; d < a.dim; ++d) wt *= np[d] == 2 ? 0.5 : 1.0;
          int q = cnt++;  // insertion by ascending coarse DoF
          while (q > 0 && c[q - 1] > cd) { c[q] = c[q - 1]; w[q] = w[q - 1]; --q; }
          c[q] = cd; w[q] = wt;
        }
    if constexpr (!FILL) {
      a.rowptr[f] = cnt;
    } else {
      const int32_t o = a.rowptr[f];
      for (int q = 0; q < cnt; ++q) { a.col[o + q] = c[q]; a.val[o + q] = w[q]; }
    }
  }
}

// Row c of P^T (column c of P): the fine vertices around the coarse vertex, ascending fine DoF.
template <bool FILL>
__global__ __launch_bounds__(kThreads) void tr_restriction_kernel(TransferArgs a) {
  for (int64_t ci = (int64_t)blockIdx.x * kThreads + threadIdx.x; ci < a.n_coarse; ci += (int64_t)gridDim.x * kThreads) {
    int cnt = 0;
    int32_t r[27];
    double w[27];
    if (!a.coarse_boundary[ci]) {
      unsigned long long v[3];
      tr_unpack(a.coarse_vertex[ci], v);
      const int nz = a.dim == 3 ? 3 : 1;
      for (int iz = 0; iz < nz; ++iz)
        for (int iy = 0; iy < 3; ++iy)
          for (int ix = 0; ix < 3; ++ix) {
            const int d[3] = {ix - 1, iy - 1, a.dim == 3 ? iz - 1 : 0};
            unsigned long long p[3];
            bool inside = true;
            for (int e = 0; e < 3; ++e) {
              if (d[e] < 0 && v[e] < a.half) inside = false;
              p[e] = d[e] < 0 ? v[e] - a.half : d[e] > 0 ? v[e] + a.half : v[e];
              if (p[e] > 0x1FFFFF) inside = false;
            }
            if (!inside) continue;
            const int32_t fd = tr_lookup(a.fkeys, a.fdof, a.fmask, tr_pack(p));
            if (fd < 0) continue;
            double wt = 1.0;
            for (int e = 0; e < a.dim; ++e) wt *= d[e] != 0 ? 0.5 : 1.0;
            int q = cnt++;
            while (q > 0 && r[q - 1] > fd) { r[q] = r[q - 1]; w[q] = w[q - 1]; --q; }
            r[q] = fd; w[q] = wt;
          }
    }
    if constexpr (!FILL) {
      a.rowptr[ci] = cnt;
    } else {
      const int32_t o = a.rowptr[ci];
      for (int q = 0; q < cnt; ++q) { a.col[o + q] = r[q]; a.val[o + q] = w[q]; }
    }
  }
}

// p[0 .. n] <- exclusive scan of p[0 .. n) by one workgroup of 1024 threads, 16 consecutive entries per thread and pass
// (level vectors have 10^4 .. 10^6 entries: ~100 passes for the 121^3 lattice, < 1 ms)
__global__ __launch_bounds__(1024) void tr_scan_kernel(int32_t *p, int64_t n) {
  __shared__ int32_t part[1024];
  __shared__ int32_t carry_s;
  constexpr int kPer = 16;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < n; base += 1024 * kPer) {
    const int64_t i0 = base + (int64_t)threadIdx.x * kPer;
    int32_t v[kPer];
    int32_t sum = 0;
#pragma unroll
    for (int j = 0; j < kPer; ++j) { v[j] = i0 + j < n ? p[i0 + j] : 0; sum += v[j]; }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan of the threads' sums
      const int32_t t = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0;
      __syncthreads();
      part[threadIdx.x] += t;
      __syncthreads();
    }
    int32_t run = carry_s + part[threadIdx.x] - sum;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      if (i0 + j < n) p[i0 + j] = run;
      run += v[j];
    }
    __syncthreads();
    if (threadIdx.x == 1023) carry_s += part[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) p[n] = carry_s;
}

}  // namespace gmg
