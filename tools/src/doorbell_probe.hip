// developer probe: can a resident kernel be fed through hipStreamWriteValue64 doorbells in host-mapped memory, and what
// does one hand-off cost? (decides the transport of the persistent block mode)   hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Ctrl { unsigned long long ring[256]; unsigned done[1024]; unsigned long long t_seen[256]; };
// variant 2: the ring lives in DEVICE memory (written by hipStreamWriteValue64), progress goes to host-mapped memory
__global__ void poller2(unsigned long long* ring, Ctrl* c, int n_cmds, long long max_spin)
{
  __shared__ unsigned long long cmd;
  for (int k = 0; k < n_cmds; k++)
  {
    if (threadIdx.x == 0)
    {
      long long spins = 0;
      unsigned long long v;
      do {
        v = __hip_atomic_load(&ring[k & 255], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((unsigned)(v >> 32) == (unsigned)(k + 1)) break;
        __builtin_amdgcn_s_sleep(8);
      } while (++spins < max_spin);
      cmd = ((unsigned)(v >> 32) == (unsigned)(k + 1)) ? v : ~0ull;
    }
    __syncthreads();
    if (cmd == ~0ull) break;
    __syncthreads();
    if (threadIdx.x == 0 && ((k & 15) == 15 || k == n_cmds - 1))
      __hip_atomic_store(&c->done[blockIdx.x], (unsigned)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__global__ void poller(Ctrl* c, int n_cmds, long long max_spin)
{
  __shared__ unsigned long long cmd;
  for (int k = 0; k < n_cmds; k++)
  {
    if (threadIdx.x == 0)
    {
      long long spins = 0;
      unsigned long long v;
      do {
        v = __hip_atomic_load(&c->ring[k & 255], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((unsigned)(v >> 32) == (unsigned)(k + 1)) break;
        __builtin_amdgcn_s_sleep(8);
      } while (++spins < max_spin);
      cmd = ((unsigned)(v >> 32) == (unsigned)(k + 1)) ? v : ~0ull;
      if (blockIdx.x == 0) c->t_seen[k & 255] = wall_clock64();
    }
    __syncthreads();
    if (cmd == ~0ull) break; // timed out: never hang
    __syncthreads();
    if (threadIdx.x == 0)
      __hip_atomic_store(&c->done[blockIdx.x], (unsigned)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
int main()
{
  Ctrl* h = nullptr; Ctrl* d = nullptr;
  CK(hipHostMalloc((void**)&h, sizeof(Ctrl), hipHostMallocMapped | hipHostMallocCoherent));
  memset(h, 0, sizeof(Ctrl));
  CK(hipHostGetDevicePointer((void**)&d, h, 0));
  hipStream_t sk, sd;
  CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sd, hipStreamNonBlocking));
  const int n = 200, wgs = 256;
  hipLaunchKernelGGL(poller, dim3(wgs), dim3(256), 0, sk, d, n, 1ll << 22);
  CK(hipGetLastError());
  std::this_thread::sleep_for(std::chrono::milliseconds(5));
  // (1) doorbells through stream ops, all enqueued at once
  auto t0 = std::chrono::steady_clock::now();
  int use_stream_ops = 1;
  for (int k = 0; k < n / 2; k++)
  {
    hipError_t e = hipStreamWriteValue64(sd, &d->ring[k & 255], ((unsigned long long)(k + 1) << 32) | (unsigned)k, 0);
    if (e != hipSuccess) { printf("hipStreamWriteValue64: %s -> CPU writes\n", hipGetErrorString(e)); use_stream_ops = 0; break; }
  }
  auto t1 = std::chrono::steady_clock::now();
  auto wait_done = [&](unsigned k) { long long it = 0; for (;;) { unsigned m = ~0u; for (int w = 0; w < wgs; w++) m = h->done[w] < m ? h->done[w] : m; if (m >= k) return true; if (++it > 200000000ll) return false; } };
  if (use_stream_ops)
  {
    bool ok = wait_done(n / 2);
    auto t2 = std::chrono::steady_clock::now();
    printf("stream-op doorbells: %d enqueued in %.1f us (%.2f us each), all %d WGs done %.1f us after the first enqueue: %s\n", n / 2,
           std::chrono::duration<double, std::micro>(t1 - t0).count(), std::chrono::duration<double, std::micro>(t1 - t0).count() / (n / 2), wgs,
           std::chrono::duration<double, std::micro>(t2 - t0).count(), ok ? "ok" : "TIMEOUT");
  }
  // (2) doorbells written by the CPU, one at a time, each waited for: round-trip latency host -> 256 WGs -> host
  const int k0 = use_stream_ops ? n / 2 : 0;
  double sum = 0, worst = 0;
  for (int k = k0; k < n; k++)
  {
    auto a = std::chrono::steady_clock::now();
    __atomic_store_n(&h->ring[k & 255], ((unsigned long long)(k + 1) << 32) | (unsigned)k, __ATOMIC_RELEASE);
    if (!wait_done(k + 1)) { printf("TIMEOUT at %d\n", k); break; }
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count();
    sum += us; worst = us > worst ? us : worst;
  }
  printf("CPU-written doorbells: round trip host -> all WGs -> host mean %.2f us, worst %.2f us over %d\n", sum / (n - k0), worst, n - k0);
  CK(hipStreamSynchronize(sk));
  printf("kernel exited cleanly\n");
  {
    // variant 2
    unsigned long long* dring = nullptr;
    CK(hipMalloc((void**)&dring, 256 * 8));
    CK(hipMemset(dring, 0, 256 * 8));
    memset(h, 0, sizeof(Ctrl));
    const int n2 = 2000;
    hipLaunchKernelGGL(poller2, dim3(wgs), dim3(256), 0, sk, dring, d, n2, 1ll << 22);
    CK(hipGetLastError());
    std::this_thread::sleep_for(std::chrono::milliseconds(5));
    auto a0 = std::chrono::steady_clock::now();
    for (int k = 0; k < n2; k++)
    {
      if (k >= 200 && (k & 63) == 0) // never lap the kernel by a whole ring
        while (true) { unsigned m = ~0u; for (int w = 0; w < wgs; w++) m = h->done[w] < m ? h->done[w] : m; if ((int)m + 200 > k) break; }
      CK(hipStreamWriteValue64(sd, &dring[k & 255], ((unsigned long long)(k + 1) << 32) | (unsigned)k, 0));
    }
    auto a1 = std::chrono::steady_clock::now();
    bool ok = wait_done(n2);
    auto a2 = std::chrono::steady_clock::now();
    printf("device-memory ring, stream-op doorbells: %d enqueued in %.1f us (%.2f us each); all WGs done %.1f us after the first enqueue (%.2f us per command): %s\n",
           n2, std::chrono::duration<double, std::micro>(a1 - a0).count(), std::chrono::duration<double, std::micro>(a1 - a0).count() / n2,
           std::chrono::duration<double, std::micro>(a2 - a0).count(), std::chrono::duration<double, std::micro>(a2 - a0).count() / n2, ok ? "ok" : "TIMEOUT");
    CK(hipStreamSynchronize(sk));
  }
  return 0;
}
