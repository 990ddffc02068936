// gmg_comm.hpp -- RCCL (xGMI) communicator, halo plans and scalar all-reduces.
//
// Stands in for what Epetra_MpiComm / Epetra_Import / MPI_Allreduce do underneath the
// reference's vmult and vector reductions (collective table): one
// process per GPU, the ghost values of an operator's column space are appended behind the
// locally owned entries of the vector, in neighbour order.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>

#include <vector>

#include "gmg_device.hpp"

namespace gmg {

struct Comm {
  ncclComm_t comm = nullptr;
  int rank = 0, n_ranks = 1;
  bool ready = false;
};

struct HaloPlan {
  int n_neighbors = 0;
  std::vector<int> rank, send_count, recv_count;
  int64_t total_send = 0, total_recv = 0;
  int32_t *send_idx = nullptr;  // device: owned local rows to pack, neighbour after neighbour
  double *send_buf = nullptr;   // device
};

inline void free_halo(HaloPlan &h) {
  if (h.send_idx) (void)hipFree(h.send_idx);
  if (h.send_buf) (void)hipFree(h.send_buf);
  h = HaloPlan();
}

inline int build_halo(HaloPlan &h, int n_neighbors, const int32_t *neighbor_rank, const int32_t *send_count,
                      const int32_t *send_idx, const int32_t *recv_count, hipStream_t stream) {
  h.n_neighbors = n_neighbors;
  h.total_send = h.total_recv = 0;
  for (int i = 0; i < n_neighbors; ++i) {
    h.rank.push_back(neighbor_rank[i]);
    h.send_count.push_back(send_count[i]);
    h.recv_count.push_back(recv_count[i]);
    h.total_send += send_count[i];
    h.total_recv += recv_count[i];
  }
  if (h.total_send > 0) {
    if (hipMalloc(&h.send_idx, sizeof(int32_t) * (size_t)h.total_send) != hipSuccess) return 1;
    if (hipMalloc(&h.send_buf, sizeof(double) * (size_t)h.total_send) != hipSuccess) return 1;
    if (hipMemcpyAsync(h.send_idx, send_idx, sizeof(int32_t) * (size_t)h.total_send, hipMemcpyHostToDevice, stream) != hipSuccess)
      return 1;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
  }
  return 0;
}

inline int comm_unique_id(void *out) {
  static_assert(sizeof(ncclUniqueId) <= 128, "ncclUniqueId must fit GMG_UNIQUE_ID_BYTES");
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return 1;
  memset(out, 0, 128);
  memcpy(out, &id, sizeof id);
  return 0;
}

inline int comm_init(Comm &c, int rank, int n_ranks, const void *id_bytes) {
  ncclUniqueId id;
  memcpy(&id, id_bytes, sizeof id);
  if (ncclCommInitRank(&c.comm, n_ranks, id, rank) != ncclSuccess) return 1;
  c.rank = rank; c.n_ranks = n_ranks; c.ready = true;
  return 0;
}

inline void comm_destroy(Comm &c) {
  if (c.ready && c.comm) (void)ncclCommDestroy(c.comm);
  c = Comm();
}

// x[n_owned ...] <- neighbours' owned values; all traffic on `stream`.
inline int halo_exchange(Comm &c, const HaloPlan &h, double *x, int64_t n_owned, hipStream_t stream) {
  if (h.n_neighbors == 0) return 0;
  if (!c.ready) return 1;
  if (h.total_send > 0) {
    int64_t g = (h.total_send + kThreads - 1) / kThreads;
    if (g > kMaxPartials) g = kMaxPartials;
    hipLaunchKernelGGL(gather_scatter_kernel, dim3((unsigned)g), dim3(kThreads), 0, stream, h.send_buf, (const int32_t *)nullptr,
                       (const double *)x, (const int32_t *)h.send_idx, h.total_send);
  }
  if (ncclGroupStart() != ncclSuccess) return 1;
  int64_t so = 0, ro = 0;
  for (int i = 0; i < h.n_neighbors; ++i) {
    if (h.send_count[(size_t)i] > 0 &&
        ncclSend(h.send_buf + so, (size_t)h.send_count[(size_t)i], ncclDouble, h.rank[(size_t)i], c.comm, stream) != ncclSuccess)
      return 1;
    if (h.recv_count[(size_t)i] > 0 &&
        ncclRecv(x + n_owned + ro, (size_t)h.recv_count[(size_t)i], ncclDouble, h.rank[(size_t)i], c.comm, stream) != ncclSuccess)
      return 1;
    so += h.send_count[(size_t)i];
    ro += h.recv_count[(size_t)i];
  }
  if (ncclGroupEnd() != ncclSuccess) return 1;
  return 0;
}

inline int allreduce_sum(Comm &c, double *dev, int count, hipStream_t stream) {
  if (!c.ready) return 1;
  return ncclAllReduce(dev, dev, (size_t)count, ncclDouble, ncclSum, c.comm, stream) != ncclSuccess;
}
inline int allreduce_max(Comm &c, double *dev, int count, hipStream_t stream) {
  if (!c.ready) return 1;
  return ncclAllReduce(dev, dev, (size_t)count, ncclDouble, ncclMax, c.comm, stream) != ncclSuccess;
}

}  // namespace gmg
