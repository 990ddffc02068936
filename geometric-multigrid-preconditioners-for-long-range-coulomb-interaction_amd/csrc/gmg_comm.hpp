// gmg_comm.hpp -- one process per GPU: halo exchange, scalar all-reduces and all-gathers between the ranks.
//
// Stands in for what Epetra_MpiComm / Epetra_Import / MPI_Allreduce do underneath the reference's vmult and vector
// reductions (collective table): the ghost values of an operator's column space are appended
// behind the locally owned entries of the vector, in neighbour order.
//
// Two transports behind the same four calls:
//   * RCCL (ncclSend/Recv groups, ncclAllReduce, ncclAllGather over xGMI) -- selected by an id from ncclGetUniqueId.
//   * PEER (GMG_COMM_TRANSPORT=peer when the id is created): every rank owns a "mailbox" in its HBM that its peers
//     map with hipIpcOpenMemHandle (on one node: peer-to-peer stores over xGMI).  A message is written straight into
//     the receiver's mailbox by a copy kernel of the sender, followed by a release store of a sequence number; the
//     receiver's kernel polls that number (system-scope acquire), unpacks, and acknowledges.  No collective launch,
//     no host synchronisation; the payload area is double-buffered by the sequence number's parity and a sender
//     waits for the acknowledgement of the message two rounds back before it reuses a buffer.  Measured
//     (tools/micro/ipc_probe.hip): 2.6 us one way between two processes.  Unlike RCCL this transport also works
//     between processes that share ONE GPU, which is how the rank-parallel layout is tested on a single-GPU box
//     (tests/test_gpu_two_ranks.py).  Every wait inside a kernel is bounded (kPeerSpinLimit polls, ~15 s): a rank that
//     gives up raises an abort flag the host reports as GMG_ERR_COMM instead of hanging the GPU.
//     Memory: everything a peer's kernel stores into -- the mailbox and the coarse CG's shared direction ring -- is
//     allocated FINE-GRAINED (hipExtMallocWithFlags, hipDeviceMallocFinegrained: coherent across devices while kernels
//     run).  Ranks on DIFFERENT devices refuse to start on coarse-grained memory (comm_init compares the PCI bus ids
//     the ranks publish in the boot segment); ranks sharing one device (the single-GPU test layout) may fall back to
//     plain hipMalloc, where the device's own L2 is the point of coherence.  GMG_PEER_COARSE=1 forces plain hipMalloc
//     (measurement of what fine-grained costs the SpMV; refused across devices).
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

constexpr int kPeerMaxRanks = 8;
constexpr char kPeerTag[] = "GMGPEER:";
constexpr int64_t kPeerFlagBytes = 16384;           // head of a mailbox: data_seq[src] at src * 128, ack_seq[dst] at 2048 + dst * 128;
                                                    // from kPeerCgOffset (4096): sums and halo tags of the coarse CG (gmg_device.hpp: PeerCG)
constexpr long long kPeerSpinLimit = 30000000LL;    // polls (~0.5 us each with a short sleep: ~15 s) after which a waiting kernel declares the exchange broken.
                                                    // The ranks meet on the host (comm_host_barrier) before a solve, so a kernel never waits out a peer's host work.

// start-up only: a POSIX shared-memory object carries the IPC handles and a host barrier
struct PeerBoot {  // zero-filled by ftruncate
  std::atomic<int> arrived, generation;
  int64_t cap_bytes;  // payload capacity per (parity, source)
  hipIpcMemHandle_t handle[kPeerMaxRanks];
  hipIpcMemHandle_t shared_handle[kPeerMaxRanks];  // comm_share_alloc: one collective allocation at a time
  int64_t meta[kPeerMaxRanks][4 + kPeerMaxRanks];  // comm_exchange_meta
  int64_t dev_uid[kPeerMaxRanks];                  // PCI domain / bus / device of every rank's GPU (same value: the ranks share it)
  int32_t fine[kPeerMaxRanks];                     // 1: that rank's mailbox is fine-grained memory
};

struct Comm {
  ncclComm_t comm = nullptr;
  int rank = 0, n_ranks = 1;
  bool ready = false;
  // peer transport
  bool peer = false;
  PeerBoot *boot = nullptr;
  char boot_name[96] = {};
  char *box[kPeerMaxRanks] = {};  // device: my mailbox (box[rank]) and the peers' (IPC-mapped)
  int64_t cap = 0;                // bytes per (parity, source)
  unsigned long long seq = 0;     // collective rounds so far (the same on every rank: collectives are called in the same order)
  unsigned long long last_sent[kPeerMaxRanks][2] = {};  // round of my last message to a peer, per parity
  unsigned int *cnt = nullptr;    // device: workgroups done, per peer (last one publishes)
  int *abort_host = nullptr;      // pinned: set by a kernel that gave up waiting
  bool box_fine = false;          // my mailbox is fine-grained memory
  int ring_fine = -1;             // the shared direction ring: -1 not allocated, 0 plain hipMalloc, 1 fine-grained
  int n_devices = 1;              // distinct GPUs under the ranks
  char why[160] = {};             // text of the last start-up failure
};

// memory a peer's kernels store into while my kernels run: fine-grained where the runtime offers it
inline int peer_alloc(void **p, size_t bytes, bool *fine) {
  const char *coarse = std::getenv("GMG_PEER_COARSE");
  *fine = false;
  if (!(coarse && coarse[0] == '1')) {
    if (hipExtMallocWithFlags(p, bytes, hipDeviceMallocFinegrained) == hipSuccess) { *fine = true; return 0; }
    (void)hipGetLastError();
  }
  return hipMalloc(p, bytes) != hipSuccess;
}

inline int boot_barrier(PeerBoot *b, int n_ranks) {
  const int gen = b->generation.load(std::memory_order_acquire);
  if (b->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == n_ranks) {
    b->arrived.store(0, std::memory_order_relaxed);
    b->generation.fetch_add(1, std::memory_order_release);
    return 0;
  }
  const auto t0 = std::chrono::steady_clock::now();
  long spins = 0;
  while (b->generation.load(std::memory_order_acquire) == gen) {
    if ((++spins & 0xfff) == 0) {
      (void)sched_yield();
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return 1;
    }
  }
  return 0;
}

struct HaloPlan {
  bool active = false;  // a plan was set for this operator (possibly with no neighbours: the rank still joins the round)
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
  h.active = true;
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
  if (tr && std::strcmp(tr, "peer") == 0) {
    // the id names a fresh shared-memory object used at start-up only (IPC handles + a host barrier)
    const char *mb = std::getenv("GMG_PEER_SLOT_MB");
    char name[80];
    std::snprintf(name, sizeof name, "/gmgpeer_%d_%lld", (int)getpid(),
                  (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return 1;
    if (ftruncate(fd, (off_t)sizeof(PeerBoot)) != 0) { close(fd); shm_unlink(name); return 1; }
    void *p = mmap(nullptr, sizeof(PeerBoot), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { shm_unlink(name); return 1; }
    static_cast<PeerBoot *>(p)->cap_bytes = (int64_t)(mb ? std::atoi(mb) : 32) << 20;
    munmap(p, sizeof(PeerBoot));
    memset(out, 0, 128);
    std::snprintf(static_cast<char *>(out), 128, "%s%s", kPeerTag, name);
    return 0;
  }
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return 1;
  memset(out, 0, 128);
  memcpy(out, &id, sizeof id);
  return 0;
}

inline int comm_init(Comm &c, int rank, int n_ranks, const void *id_bytes) {
  if (std::memcmp(id_bytes, kPeerTag, sizeof(kPeerTag) - 1) == 0) {
    if (n_ranks > kPeerMaxRanks) return 1;
    std::snprintf(c.boot_name, sizeof c.boot_name, "%s", static_cast<const char *>(id_bytes) + sizeof(kPeerTag) - 1);
    const int fd = shm_open(c.boot_name, O_RDWR, 0600);
    if (fd < 0) return 1;
    void *p = mmap(nullptr, sizeof(PeerBoot), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return 1;
    c.boot = static_cast<PeerBoot *>(p);
    c.cap = c.boot->cap_bytes;
    c.rank = rank; c.n_ranks = n_ranks;
    const size_t bytes = (size_t)kPeerFlagBytes + 2 * (size_t)n_ranks * (size_t)c.cap;
    // The mailbox is polled by kernels of this GPU while kernels of OTHER GPUs store into it: fine-grained memory.
    if (peer_alloc((void **)&c.box[rank], bytes, &c.box_fine)) return 1;
    {
      int dev = 0;
      char bus[64] = {};
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetPCIBusId(bus, sizeof bus, dev) != hipSuccess) return 1;
      int64_t uid = 1469598103934665603ll;  // FNV-1a of the bus id string ("0000:c1:00.0")
      for (const char *q = bus; *q; ++q) uid = (uid ^ (unsigned char)*q) * 1099511628211ll;
      c.boot->dev_uid[rank] = uid;
      c.boot->fine[rank] = c.box_fine ? 1 : 0;
    }
    if (hipMemset(c.box[rank], 0, (size_t)kPeerFlagBytes) != hipSuccess) return 1;
    if (hipMalloc((void **)&c.cnt, sizeof(unsigned int) * 2 * kPeerMaxRanks) != hipSuccess) return 1;
    if (hipMemset(c.cnt, 0, sizeof(unsigned int) * 2 * kPeerMaxRanks) != hipSuccess) return 1;
    if (hipHostMalloc((void **)&c.abort_host, sizeof(int), hipHostMallocDefault) != hipSuccess) return 1;
    *c.abort_host = 0;
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipIpcGetMemHandle(&c.boot->handle[rank], c.box[rank]) != hipSuccess) return 1;
    if (boot_barrier(c.boot, n_ranks)) return 1;
    {
      // every rank sees every rank's device and memory type: the same verdict everywhere, before anything is mapped
      c.n_devices = 0;
      bool all_fine = true;
      for (int r = 0; r < n_ranks; ++r) {
        bool seen = false;
        for (int q = 0; q < r; ++q) seen = seen || c.boot->dev_uid[q] == c.boot->dev_uid[r];
        c.n_devices += seen ? 0 : 1;
        all_fine = all_fine && c.boot->fine[r] != 0;
      }
      if (c.n_devices > 1 && !all_fine) {
        std::snprintf(c.why, sizeof c.why, "peer transport: %d ranks on %d devices but a mailbox is coarse-grained memory (no cross-device visibility while kernels run)",
                      n_ranks, c.n_devices);
        std::fprintf(stderr, "[gmg] %s\n", c.why);
        return 1;
      }
    }
    for (int r = 0; r < n_ranks; ++r)
      if (r != rank && hipIpcOpenMemHandle((void **)&c.box[r], c.boot->handle[r], hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
        std::snprintf(c.why, sizeof c.why, "peer transport: hipIpcOpenMemHandle of rank %d's mailbox failed", r);
        return 1;
      }
    c.peer = true; c.ready = true;
    return boot_barrier(c.boot, n_ranks);
  }
  ncclUniqueId id;
  memcpy(&id, id_bytes, sizeof id);
  if (ncclCommInitRank(&c.comm, n_ranks, id, rank) != ncclSuccess) return 1;
  c.rank = rank; c.n_ranks = n_ranks; c.ready = true;
  c.n_devices = n_ranks;  // (RCCL refuses two ranks on one device)
  return 0;
}

inline void comm_destroy(Comm &c) {
  if (c.ready && c.peer) {
    (void)hipDeviceSynchronize();
    (void)boot_barrier(c.boot, c.n_ranks);  // nobody unmaps a mailbox a peer may still write to
    for (int r = 0; r < c.n_ranks; ++r)
      if (r != c.rank && c.box[r]) (void)hipIpcCloseMemHandle(c.box[r]);
    (void)boot_barrier(c.boot, c.n_ranks);
    if (c.box[c.rank]) (void)hipFree(c.box[c.rank]);
    if (c.cnt) (void)hipFree(c.cnt);
    if (c.abort_host) (void)hipHostFree(c.abort_host);
    if (c.rank == 0) (void)shm_unlink(c.boot_name);
    (void)munmap(c.boot, sizeof(PeerBoot));
  } else if (c.ready && c.comm) {
    (void)ncclCommDestroy(c.comm);
  }
  c = Comm();
}

// ---------------------------------------------------------------- peer transport: kernels

struct PeerSendArgs {
  char *peer_box[kPeerMaxRanks];       // mailbox of the i-th destination
  const double *src[kPeerMaxRanks];
  long long count[kPeerMaxRanks];      // doubles
  unsigned long long wait_ack[kPeerMaxRanks];  // round whose acknowledgement frees the buffer (0: none)
  int dst_rank[kPeerMaxRanks];
  int n, me, n_ranks;
  unsigned long long seq;
  long long cap;
  char *my_box;
  unsigned int *cnt;
  int *abort_flag;
};

__device__ __forceinline__ unsigned long long sys_load(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void sys_store(unsigned long long *p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// true when *flag >= want arrived in time; false (and the abort flag raised / seen) otherwise.  One thread polls.
__device__ __forceinline__ bool peer_wait(const unsigned long long *flag, unsigned long long want, int *abort_flag) {
  __shared__ int ok_s;
  if (threadIdx.x == 0) {
    int ok = 1;
    // relaxed polls (an acquire load would invalidate the caches on every poll); one acquire fence at the end
    for (long long spins = 0; __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want; ++spins) {
      __builtin_amdgcn_s_sleep(4);
      if ((spins & 255) == 255 && *(volatile int *)abort_flag) { ok = 0; break; }
      if (spins > kPeerSpinLimit) { *abort_flag = 1; ok = 0; break; }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    ok_s = ok;
  }
  __syncthreads();
  const bool ok = ok_s != 0;
  __syncthreads();  // (the flag may be rewritten by the next wait)
  return ok;
}

// grid (G, n): workgroups (.., i) copy src[i] into destination i's mailbox (payload[seq & 1][me]); the last one publishes seq
__global__ __launch_bounds__(kThreads) void peer_send_kernel(PeerSendArgs a) {
  const int i = blockIdx.y;
  if (*(volatile int *)a.abort_flag) return;
  if (a.wait_ack[i]) {
    const unsigned long long *ack = reinterpret_cast<const unsigned long long *>(a.my_box + 2048 + a.dst_rank[i] * 128);
    if (!peer_wait(ack, a.wait_ack[i], a.abort_flag)) return;
  }
  double *dst = reinterpret_cast<double *>(a.peer_box[i] + kPeerFlagBytes + ((long long)(a.seq & 1) * a.n_ranks + a.me) * a.cap);
  const double *src = a.src[i];
  for (long long k = (long long)blockIdx.x * kThreads + threadIdx.x; k < a.count[i]; k += (long long)gridDim.x * kThreads) dst[k] = src[k];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(&a.cnt[i], 1u) == gridDim.x - 1) {
    a.cnt[i] = 0;
    sys_store(reinterpret_cast<unsigned long long *>(a.peer_box[i] + a.me * 128), a.seq);
  }
}

struct PeerRecvArgs {
  char *peer_box[kPeerMaxRanks];   // mailbox of the i-th source (for the acknowledgement)
  double *dst[kPeerMaxRanks];
  long long count[kPeerMaxRanks];
  int src_rank[kPeerMaxRanks];
  int n, me, n_ranks;
  unsigned long long seq;
  long long cap;
  char *my_box;
  unsigned int *cnt;
  int *abort_flag;
};

// grid (G, n): wait for source i's round seq, unpack its payload, the last workgroup acknowledges
__global__ __launch_bounds__(kThreads) void peer_recv_kernel(PeerRecvArgs a) {
  const int i = blockIdx.y;
  if (*(volatile int *)a.abort_flag) return;
  const unsigned long long *flag = reinterpret_cast<const unsigned long long *>(a.my_box + a.src_rank[i] * 128);
  if (!peer_wait(flag, a.seq, a.abort_flag)) return;
  const double *src = reinterpret_cast<const double *>(a.my_box + kPeerFlagBytes + ((long long)(a.seq & 1) * a.n_ranks + a.src_rank[i]) * a.cap);
  double *dst = a.dst[i];
  for (long long k = (long long)blockIdx.x * kThreads + threadIdx.x; k < a.count[i]; k += (long long)gridDim.x * kThreads)
    dst[k] = __builtin_nontemporal_load(src + k);
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(&a.cnt[kPeerMaxRanks + i], 1u) == gridDim.x - 1) {
    a.cnt[kPeerMaxRanks + i] = 0;
    sys_store(reinterpret_cast<unsigned long long *>(a.peer_box[i] + 2048 + a.me * 128), a.seq);
  }
}

// one workgroup: dev[k] = reduction over the ranks, in rank order (every rank gets the same bits), of the values the
// peers sent (payload[seq & 1][r][k]) and the rank's own dev[k]; acknowledges every peer
__global__ __launch_bounds__(kThreads) void peer_allreduce_kernel(PeerRecvArgs a, double *dev, int count, int max_op) {
  if (*(volatile int *)a.abort_flag) return;
  for (int i = 0; i < a.n; ++i) {
    const unsigned long long *flag = reinterpret_cast<const unsigned long long *>(a.my_box + a.src_rank[i] * 128);
    if (!peer_wait(flag, a.seq, a.abort_flag)) return;
  }
  if ((int)threadIdx.x < count) {
    double acc = 0.0;
    for (int r = 0; r < a.n_ranks; ++r) {
      double v;
      if (r == a.me) v = dev[threadIdx.x];
      else v = __builtin_nontemporal_load(reinterpret_cast<const double *>(a.my_box + kPeerFlagBytes + ((long long)(a.seq & 1) * a.n_ranks + r) * a.cap) + threadIdx.x);
      acc = r == 0 ? v : (max_op ? (v > acc ? v : acc) : acc + v);
    }
    dev[threadIdx.x] = acc;
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0)
    for (int i = 0; i < a.n; ++i) sys_store(reinterpret_cast<unsigned long long *>(a.peer_box[i] + 2048 + a.me * 128), a.seq);
}

// ---------------------------------------------------------------- peer transport: host side

struct PeerMsg { int peer; const double *src; double *dst; int64_t send_count, recv_count; };

inline int peer_grid(int64_t count) { return (int)std::max<int64_t>(1, std::min<int64_t>(64, (count + 4095) / 4096)); }

// one round: send msgs[i].src (send_count doubles) to msgs[i].peer and receive recv_count doubles from it into msgs[i].dst
// (either count may be 0).  Every rank calls the collectives in the same order, so the round number needs no negotiation.
inline int peer_exchange(Comm &c, const std::vector<PeerMsg> &msgs, hipStream_t stream, bool unpack = true) {
  const unsigned long long seq = ++c.seq;
  PeerSendArgs s{};
  PeerRecvArgs r{};
  int max_send_grid = 1, max_recv_grid = 1;
  for (const PeerMsg &m : msgs) {
    if ((int64_t)sizeof(double) * std::max(m.send_count, m.recv_count) > c.cap) return 1;  // GMG_PEER_SLOT_MB too small
    if (m.send_count > 0) {
      const int i = s.n++;
      s.peer_box[i] = c.box[m.peer]; s.src[i] = m.src; s.count[i] = m.send_count; s.dst_rank[i] = m.peer;
      s.wait_ack[i] = c.last_sent[m.peer][seq & 1];
      c.last_sent[m.peer][seq & 1] = seq;
      max_send_grid = std::max(max_send_grid, peer_grid(m.send_count));
    }
    if (m.recv_count > 0) {
      const int i = r.n++;
      r.peer_box[i] = c.box[m.peer]; r.dst[i] = m.dst; r.count[i] = m.recv_count; r.src_rank[i] = m.peer;
      max_recv_grid = std::max(max_recv_grid, peer_grid(m.recv_count));
    }
  }
  s.me = r.me = c.rank; s.n_ranks = r.n_ranks = c.n_ranks; s.seq = r.seq = seq; s.cap = r.cap = c.cap;
  s.my_box = r.my_box = c.box[c.rank]; s.cnt = r.cnt = c.cnt; s.abort_flag = r.abort_flag = c.abort_host;
  if (s.n) hipLaunchKernelGGL(peer_send_kernel, dim3(max_send_grid, s.n), dim3(kThreads), 0, stream, s);
  if (r.n && unpack) hipLaunchKernelGGL(peer_recv_kernel, dim3(max_recv_grid, r.n), dim3(kThreads), 0, stream, r);
  return hipGetLastError() != hipSuccess;
}

inline int peer_allreduce(Comm &c, double *dev, int count, bool max_op, hipStream_t stream) {
  if (count > kThreads) return 1;
  std::vector<PeerMsg> msgs;
  for (int p = 0; p < c.n_ranks; ++p)
    if (p != c.rank) msgs.push_back(PeerMsg{p, dev, nullptr, count, count});
  if (peer_exchange(c, msgs, stream, false)) return 1;  // sends only; the reduce kernel below receives
  PeerRecvArgs r{};
  for (const PeerMsg &m : msgs) { const int i = r.n++; r.peer_box[i] = c.box[m.peer]; r.src_rank[i] = m.peer; }
  r.me = c.rank; r.n_ranks = c.n_ranks; r.seq = c.seq; r.cap = c.cap; r.my_box = c.box[c.rank]; r.cnt = c.cnt; r.abort_flag = c.abort_host;
  // the peers' copies of MY value leave from dev before the reduce kernel overwrites it: same stream, in order
  hipLaunchKernelGGL(peer_allreduce_kernel, dim3(1), dim3(kThreads), 0, stream, r, dev, count, max_op ? 1 : 0);
  return hipGetLastError() != hipSuccess;
}

// ---------------------------------------------------------------- the four calls

// x[n_owned ...] <- neighbours' owned values; all traffic on `stream`.
inline int halo_exchange(Comm &c, const HaloPlan &h, double *x, int64_t n_owned, hipStream_t stream) {
  if (h.n_neighbors == 0 && !c.peer) return 0;
  if (!c.ready) return 1;
  if (h.total_send > 0) {
    int64_t g = (h.total_send + kThreads - 1) / kThreads;
    if (g > kMaxPartials) g = kMaxPartials;
    hipLaunchKernelGGL(gather_scatter_kernel, dim3((unsigned)g), dim3(kThreads), 0, stream, h.send_buf, (const int32_t *)nullptr,
                       (const double *)x, (const int32_t *)h.send_idx, h.total_send);
  }
  if (c.peer) {  // (a rank without neighbours still takes part in the round: the round numbers stay in step)
    std::vector<PeerMsg> msgs;
    int64_t so = 0, ro = 0;
    for (int i = 0; i < h.n_neighbors; ++i) {
      msgs.push_back(PeerMsg{h.rank[(size_t)i], h.send_buf + so, x + n_owned + ro, h.send_count[(size_t)i], h.recv_count[(size_t)i]});
      so += h.send_count[(size_t)i];
      ro += h.recv_count[(size_t)i];
    }
    return peer_exchange(c, msgs, stream);
  }
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
  if (c.peer) return peer_allreduce(c, dev, count, false, stream);
  return ncclAllReduce(dev, dev, (size_t)count, ncclDouble, ncclSum, c.comm, stream) != ncclSuccess;
}
inline int allreduce_max(Comm &c, double *dev, int count, hipStream_t stream) {
  if (!c.ready) return 1;
  if (c.peer) return peer_allreduce(c, dev, count, true, stream);
  return ncclAllReduce(dev, dev, (size_t)count, ncclDouble, ncclMax, c.comm, stream) != ncclSuccess;
}

// full[r * chunk ...) <- rank r's chunk (own chunk already in place), for every r
inline int allgather_chunks(Comm &c, double *full, int64_t chunk, hipStream_t stream) {
  if (!c.ready) return 1;
  if (c.peer) {
    std::vector<PeerMsg> msgs;
    for (int p = 0; p < c.n_ranks; ++p)
      if (p != c.rank) msgs.push_back(PeerMsg{p, full + (int64_t)c.rank * chunk, full + (int64_t)p * chunk, chunk, chunk});
    return peer_exchange(c, msgs, stream);
  }
  return ncclAllGather(full + (int64_t)c.rank * chunk, full, (size_t)chunk, ncclDouble, c.comm, stream) != ncclSuccess;
}

// Collective: every rank allocates `bytes` of device memory and maps everybody else's allocation (the coarse CG keeps
// its direction vectors there: the neighbours write their halo entries straight into them).  local[rank] is the own one.
inline int comm_share_alloc(Comm &c, size_t bytes, char *ptrs[kPeerMaxRanks]) {
  if (!c.peer) return 1;
  for (int r = 0; r < kPeerMaxRanks; ++r) ptrs[r] = nullptr;
  // the neighbours' kernels store into it (halo entries of d) while mine run: fine-grained like the mailbox; across devices
  // nothing else is accepted
  bool fine = false;
  if (peer_alloc((void **)&ptrs[c.rank], bytes, &fine)) return 1;
  c.ring_fine = fine ? 1 : 0;
  if (c.n_devices > 1 && !fine) {
    std::snprintf(c.why, sizeof c.why, "peer transport: shared vectors on %d devices need fine-grained memory", c.n_devices);
    std::fprintf(stderr, "[gmg] %s\n", c.why);
    return 1;
  }
  if (hipMemset(ptrs[c.rank], 0, bytes) != hipSuccess) return 1;
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (hipIpcGetMemHandle(&c.boot->shared_handle[c.rank], ptrs[c.rank]) != hipSuccess) return 1;
  if (boot_barrier(c.boot, c.n_ranks)) return 1;
  for (int r = 0; r < c.n_ranks; ++r)
    if (r != c.rank && hipIpcOpenMemHandle((void **)&ptrs[r], c.boot->shared_handle[r], hipIpcMemLazyEnablePeerAccess) != hipSuccess) return 1;
  return boot_barrier(c.boot, c.n_ranks);  // (the handle table may be reused after this)
}
inline void comm_share_free(Comm &c, char *ptrs[kPeerMaxRanks]) {
  if (!c.peer || !ptrs[c.rank]) return;
  (void)hipDeviceSynchronize();
  (void)boot_barrier(c.boot, c.n_ranks);  // nobody still writes into a vector about to be unmapped
  for (int r = 0; r < c.n_ranks; ++r)
    if (r != c.rank && ptrs[r]) (void)hipIpcCloseMemHandle(ptrs[r]);
  (void)boot_barrier(c.boot, c.n_ranks);
  (void)hipFree(ptrs[c.rank]);
  for (int r = 0; r < kPeerMaxRanks; ++r) ptrs[r] = nullptr;
}
// Collective: every rank publishes a few integers, everybody reads everybody's.
inline int comm_exchange_meta(Comm &c, const int64_t *mine, int n, int64_t all[kPeerMaxRanks][4 + kPeerMaxRanks]) {
  if (!c.peer || n > 4 + kPeerMaxRanks) return 1;
  for (int k = 0; k < n; ++k) c.boot->meta[c.rank][k] = mine[k];
  if (boot_barrier(c.boot, c.n_ranks)) return 1;
  for (int r = 0; r < c.n_ranks; ++r)
    for (int k = 0; k < n; ++k) all[r][k] = c.boot->meta[r][k];
  return boot_barrier(c.boot, c.n_ranks);
}

// The ranks meet on the host: called before a solve, so that a rank that finished its host-side setup early does not
// park a waiting kernel on its GPU for seconds.  (RCCL: its collectives are enqueued, not spun on: nothing to do.)
inline int comm_host_barrier(Comm &c) {
  if (!c.ready || !c.peer) return 0;
  return boot_barrier(c.boot, c.n_ranks);
}

inline bool comm_aborted(const Comm &c) { return c.peer && c.abort_host && *c.abort_host != 0; }

}  // namespace gmg
