// gmg_comm.hpp -- RCCL (xGMI) communicator, halo plans and scalar all-reduces.
//
// Two transports behind the same calls: RCCL (the product path: one process per GPU over xGMI)
// and, selected by GMG_COMM_TRANSPORT=shm when the id is created, host-staged POSIX shared
// memory for ranks of one node -- slow (every call synchronises the stream), but it lets two
// processes share ONE GPU, which RCCL refuses, so the rank-parallel layout (halo pack/unpack,
// partitioned level 0, all-gathers) is exercised end to end on a single-GPU box
// (tests/test_gpu_two_ranks.py).
//
// Stands in for what Epetra_MpiComm / Epetra_Import / MPI_Allreduce do underneath the
// reference's vmult and vector reductions (collective table): one
// process per GPU, the ghost values of an operator's column space are appended behind the
// locally owned entries of the vector, in neighbour order.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <fcntl.h>
#include <stdint.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gmg_device.hpp"

namespace gmg {

constexpr int kShmMaxRanks = 8;
constexpr char kShmTag[] = "GMGSHM:";

struct ShmHeader {  // zero-filled by ftruncate
  std::atomic<int> arrived, generation;
  int64_t slot_bytes;
  int64_t seg_off[kShmMaxRanks][kShmMaxRanks];  // [sender][receiver]: offset (doubles) of the halo segment in the sender's slot
  int64_t seg_cnt[kShmMaxRanks][kShmMaxRanks];
};

struct Comm {
  ncclComm_t comm = nullptr;
  int rank = 0, n_ranks = 1;
  bool ready = false;
  // shared-memory transport
  bool shm = false;
  ShmHeader *hdr = nullptr;
  char *slots = nullptr;
  size_t map_bytes = 0;
  char shm_name[96] = {};
  double *slot(int r) const { return reinterpret_cast<double *>(slots + (size_t)r * (size_t)hdr->slot_bytes); }
};

inline size_t shm_total_bytes(int64_t slot_bytes) { return 8192 + (size_t)kShmMaxRanks * (size_t)slot_bytes; }

// every rank calls this the same number of times; gives up after 5 minutes (a peer died)
inline int shm_barrier(Comm &c) {
  const int gen = c.hdr->generation.load(std::memory_order_acquire);
  if (c.hdr->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c.n_ranks) {
    c.hdr->arrived.store(0, std::memory_order_relaxed);
    c.hdr->generation.fetch_add(1, std::memory_order_release);
    return 0;
  }
  const auto t0 = std::chrono::steady_clock::now();
  long spins = 0;
  while (c.hdr->generation.load(std::memory_order_acquire) == gen) {
    if ((++spins & 0xfff) == 0) {
      (void)sched_yield();
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(300)) return 1;
    }
  }
  return 0;
}

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
  const char *tr = std::getenv("GMG_COMM_TRANSPORT");
  if (tr && std::strcmp(tr, "shm") == 0) {
    // the id names a fresh shared-memory object: header + one slot per rank
    const char *mb = std::getenv("GMG_SHM_SLOT_MB");
    const int64_t slot_bytes = (int64_t)(mb ? std::atoi(mb) : 16) << 20;
    char name[80];
    std::snprintf(name, sizeof name, "/gmgshm_%d_%lld", (int)getpid(),
                  (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return 1;
    if (ftruncate(fd, (off_t)shm_total_bytes(slot_bytes)) != 0) { close(fd); shm_unlink(name); return 1; }
    void *p = mmap(nullptr, sizeof(ShmHeader), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { shm_unlink(name); return 1; }
    static_cast<ShmHeader *>(p)->slot_bytes = slot_bytes;
    munmap(p, sizeof(ShmHeader));
    memset(out, 0, 128);
    std::snprintf(static_cast<char *>(out), 128, "%s%s", kShmTag, name);
    return 0;
  }
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return 1;
  memset(out, 0, 128);
  memcpy(out, &id, sizeof id);
  return 0;
}

inline int comm_init(Comm &c, int rank, int n_ranks, const void *id_bytes) {
  if (std::memcmp(id_bytes, kShmTag, sizeof(kShmTag) - 1) == 0) {
    static_assert(sizeof(ShmHeader) <= 8192, "header region");
    if (n_ranks > kShmMaxRanks) return 1;
    std::snprintf(c.shm_name, sizeof c.shm_name, "%s", static_cast<const char *>(id_bytes) + sizeof(kShmTag) - 1);
    const int fd = shm_open(c.shm_name, O_RDWR, 0600);
    if (fd < 0) return 1;
    void *h = mmap(nullptr, sizeof(ShmHeader), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (h == MAP_FAILED) { close(fd); return 1; }
    const int64_t slot_bytes = static_cast<ShmHeader *>(h)->slot_bytes;
    munmap(h, sizeof(ShmHeader));
    c.map_bytes = shm_total_bytes(slot_bytes);
    void *p = mmap(nullptr, c.map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return 1;
    c.hdr = static_cast<ShmHeader *>(p);
    c.slots = static_cast<char *>(p) + 8192;
    c.rank = rank; c.n_ranks = n_ranks; c.shm = true; c.ready = true;
    return shm_barrier(c);
  }
  ncclUniqueId id;
  memcpy(&id, id_bytes, sizeof id);
  if (ncclCommInitRank(&c.comm, n_ranks, id, rank) != ncclSuccess) return 1;
  c.rank = rank; c.n_ranks = n_ranks; c.ready = true;
  return 0;
}

inline void comm_destroy(Comm &c) {
  if (c.ready && c.shm) {
    if (c.rank == 0) (void)shm_unlink(c.shm_name);
    (void)munmap(c.hdr, c.map_bytes);
  } else if (c.ready && c.comm) {
    (void)ncclCommDestroy(c.comm);
  }
  c = Comm();
}

// ---- shared-memory transport: device -> own slot, barrier, peers' slots -> device, barrier
inline int shm_halo_exchange(Comm &c, const HaloPlan &h, double *x, int64_t n_owned, hipStream_t stream) {
  if ((int64_t)sizeof(double) * h.total_send > c.hdr->slot_bytes) return 1;
  for (int r = 0; r < c.n_ranks; ++r) c.hdr->seg_cnt[c.rank][r] = 0;
  int64_t so = 0;
  for (int i = 0; i < h.n_neighbors; ++i) {
    c.hdr->seg_off[c.rank][h.rank[(size_t)i]] = so;
    c.hdr->seg_cnt[c.rank][h.rank[(size_t)i]] = h.send_count[(size_t)i];
    so += h.send_count[(size_t)i];
  }
  if (h.total_send > 0 &&
      hipMemcpyAsync(c.slot(c.rank), h.send_buf, sizeof(double) * (size_t)h.total_send, hipMemcpyDeviceToHost, stream) != hipSuccess)
    return 1;
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;
  if (shm_barrier(c)) return 1;
  int64_t ro = 0;
  for (int i = 0; i < h.n_neighbors; ++i) {
    const int peer = h.rank[(size_t)i];
    const int64_t cnt = h.recv_count[(size_t)i];
    if (cnt > 0) {
      if (c.hdr->seg_cnt[peer][c.rank] != cnt) return 1;  // the two halo plans disagree
      if (hipMemcpyAsync(x + n_owned + ro, c.slot(peer) + c.hdr->seg_off[peer][c.rank], sizeof(double) * (size_t)cnt,
                         hipMemcpyHostToDevice, stream) != hipSuccess)
        return 1;
    }
    ro += cnt;
  }
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;
  return shm_barrier(c);
}

inline int shm_allreduce(Comm &c, double *dev, int count, bool max_op, hipStream_t stream) {
  if (count > 64) return 1;
  double mine[64], acc[64];
  if (hipMemcpyAsync(mine, dev, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, stream) != hipSuccess) return 1;
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;
  std::memcpy(c.slot(c.rank), mine, sizeof(double) * (size_t)count);
  if (shm_barrier(c)) return 1;
  for (int k = 0; k < count; ++k) {  // rank order: the same result on every rank
    double a = c.slot(0)[k];
    for (int r = 1; r < c.n_ranks; ++r) {
      const double v = c.slot(r)[k];
      a = max_op ? (v > a ? v : a) : a + v;
    }
    acc[k] = a;
  }
  if (hipMemcpyAsync(dev, acc, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, stream) != hipSuccess) return 1;
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;
  return shm_barrier(c);
}

// full[r * chunk ...) <- rank r's chunk (own chunk already in place)
inline int shm_allgather(Comm &c, double *full, int64_t chunk, hipStream_t stream) {
  if ((int64_t)sizeof(double) * chunk > c.hdr->slot_bytes) return 1;
  if (hipMemcpyAsync(c.slot(c.rank), full + (int64_t)c.rank * chunk, sizeof(double) * (size_t)chunk, hipMemcpyDeviceToHost, stream) !=
      hipSuccess)
    return 1;
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;
  if (shm_barrier(c)) return 1;
  for (int r = 0; r < c.n_ranks; ++r)
    if (r != c.rank && hipMemcpyAsync(full + (int64_t)r * chunk, c.slot(r), sizeof(double) * (size_t)chunk, hipMemcpyHostToDevice,
                                      stream) != hipSuccess)
      return 1;
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;
  return shm_barrier(c);
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
  if (c.shm) return shm_halo_exchange(c, h, x, n_owned, stream);
  if (ncclGroupStart() != ncclSuccess) return 1;
  int64_t so = 0, ro = 0;
  bool failed = false;
  for (int i = 0; i < h.n_neighbors && !failed; ++i) {
    if (h.send_count[(size_t)i] > 0)
      failed = ncclSend(h.send_buf + so, (size_t)h.send_count[(size_t)i], ncclDouble, h.rank[(size_t)i], c.comm, stream) != ncclSuccess;
    if (!failed && h.recv_count[(size_t)i] > 0)
      failed = ncclRecv(x + n_owned + ro, (size_t)h.recv_count[(size_t)i], ncclDouble, h.rank[(size_t)i], c.comm, stream) != ncclSuccess;
    so += h.send_count[(size_t)i];
    ro += h.recv_count[(size_t)i];
  }
  if (ncclGroupEnd() != ncclSuccess) return 1;  // always closed, also after a failed send / recv
  return failed ? 1 : 0;
}

inline int allreduce_sum(Comm &c, double *dev, int count, hipStream_t stream) {
  if (!c.ready) return 1;
  if (c.shm) return shm_allreduce(c, dev, count, false, stream);
  return ncclAllReduce(dev, dev, (size_t)count, ncclDouble, ncclSum, c.comm, stream) != ncclSuccess;
}
inline int allreduce_max(Comm &c, double *dev, int count, hipStream_t stream) {
  if (!c.ready) return 1;
  if (c.shm) return shm_allreduce(c, dev, count, true, stream);
  return ncclAllReduce(dev, dev, (size_t)count, ncclDouble, ncclMax, c.comm, stream) != ncclSuccess;
}

}  // namespace gmg
