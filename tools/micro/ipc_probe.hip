// Feasibility probe: two processes on ONE GPU exchange device memory through hipIpcGetMemHandle / hipIpcOpenMemHandle
// and hand a flag back and forth from inside kernels (bounded spins).  Prints the round-trip time.
// hipcc -O3 --offload-arch=gfx950 ipc_probe.hip -o ipc_probe && ./ipc_probe
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>

struct Shared {
  std::atomic<int> arrived;
  hipIpcMemHandle_t handle[2];
  int ok[2];
};

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "rank %d: %s -> %s\n", rank, #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void pingpong(unsigned long long *mine, unsigned long long *peer, double *peer_data, const double *my_data, int rank, int rounds, int *err) {
  // rank 0 writes round r (odd steps) into the peer's flag and waits for the echo
  for (int r = 1; r <= rounds; ++r) {
    if (rank == 0) {
      peer_data[threadIdx.x] = (double)r;
      __threadfence_system();
      if (threadIdx.x == 0) __hip_atomic_store(peer, (unsigned long long)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (threadIdx.x == 0) {
      long spins = 0;
      while (__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned long long)r) {
        if (++spins > 200000000L) { *err = 1; break; }
      }
    }
    __syncthreads();
    if (*err) return;
    if (my_data[threadIdx.x] != (double)r && rank == 1) { *err = 2; return; }
    if (rank == 1) {
      peer_data[threadIdx.x] = (double)r;
      __threadfence_system();
      if (threadIdx.x == 0) __hip_atomic_store(peer, (unsigned long long)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
  }
}

int run(int rank, Shared *sh, unsigned flags) {
  CK(hipSetDevice(0));
  char *box = nullptr;
  if (flags) CK(hipExtMallocWithFlags((void **)&box, 1 << 20, flags));
  else CK(hipMalloc((void **)&box, 1 << 20));
  CK(hipMemset(box, 0, 1 << 20));
  CK(hipDeviceSynchronize());
  CK(hipIpcGetMemHandle(&sh->handle[rank], box));
  sh->arrived.fetch_add(1);
  while (sh->arrived.load() < 2) usleep(100);
  char *peer = nullptr;
  CK(hipIpcOpenMemHandle((void **)&peer, sh->handle[1 - rank], hipIpcMemLazyEnablePeerAccess));
  int *err;
  CK(hipHostMalloc((void **)&err, sizeof(int), 0));
  *err = 0;
  sh->arrived.fetch_add(1);
  while (sh->arrived.load() < 4) usleep(100);
  const int rounds = 2000;
  auto t0 = std::chrono::steady_clock::now();
  hipLaunchKernelGGL(pingpong, dim3(1), dim3(64), 0, 0, (unsigned long long *)box, (unsigned long long *)peer, (double *)(peer + 4096), (const double *)(box + 4096), rank, rounds, err);
  CK(hipDeviceSynchronize());
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  std::printf("rank %d flags %u: err %d, %d round trips in %.1f us -> %.2f us per round trip\n", rank, flags, *err, rounds, us, us / rounds);
  sh->ok[rank] = *err == 0;
  sh->arrived.fetch_add(1);
  while (sh->arrived.load() < 6) usleep(100);
  CK(hipIpcCloseMemHandle(peer));
  CK(hipFree(box));
  return *err;
}

int main(int argc, char **argv) {
  const unsigned flags = argc > 1 ? (unsigned)std::atoi(argv[1]) : 0u;  // 0: hipMalloc; 1: hipDeviceMallocFinegrained; 3: hipDeviceMallocUncached
  Shared *sh = (Shared *)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  std::memset((void *)sh, 0, sizeof(Shared));
  const pid_t pid = fork();  // before any HIP call
  const int rank = pid == 0 ? 1 : 0;
  const int rc = run(rank, sh, flags);
  if (pid == 0) _exit(rc);
  int st = 0;
  waitpid(pid, &st, 0);
  return rc || st;
}
