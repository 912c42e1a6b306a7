// Developer probe: what a device-wide synchronize costs behind a kernel whose results the host has ALREADY seen (the tail of
// bench.py's timed region in persistent block mode: flush() sees every workgroup's "left" word, then torch.cuda.synchronize()).
// A kernel of 256 workgroups x 1,024 threads spins ~100 us, dirties `mb` MB of device memory (the rings' way back into the
// state), stores a flag into host-mapped memory and ends. Host clock: launch -> flag visible -> synchronize returns, for
//   plain      hipLaunchKernelGGL on a high-priority non-blocking stream, hipDeviceSynchronize
//   stream     ... hipStreamSynchronize of that stream only
//   ext_event  hipExtLaunchKernelGGL with a stop event (a completion signal on the dispatch itself), hipEventSynchronize
//   query      ... spinning on hipStreamQuery
//   + DeviceSync / StreamSync behind the event wait: what is left for a caller that synchronizes the device anyway
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/sync_tail tools/src/sync_tail.hip && /tmp/sync_tail
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(1024) void k(float* buf, long n_per_wg, unsigned* flag, unsigned tag, long long ticks)
{
  const long long t0 = wall_clock64();
  while ((long long)wall_clock64() - t0 < ticks) {}
  float* p = buf + (long)blockIdx.x * n_per_wg;
  for (long i = threadIdx.x; i < n_per_wg; i += blockDim.x)
    p[i] = (float)tag;
  __syncthreads();
  if (threadIdx.x == 0)
    { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __hip_atomic_store(flag + blockIdx.x, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
  const int wgs = 256;
  hipStream_t s;
  int lo, hi;
  hipDeviceGetStreamPriorityRange(&lo, &hi);
  hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi);
  unsigned* h_flag;
  hipHostMalloc((void**)&h_flag, wgs * 4, hipHostMallocMapped | hipHostMallocCoherent);
  unsigned* d_flag;
  hipHostGetDevicePointer((void**)&d_flag, h_flag, 0);
  float* buf;
  hipMalloc(&buf, 64l << 20);
  hipEvent_t ev;
  hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  unsigned tag = 0;
  for (int mb : {0, 20})
    for (int mode = 0; mode < 8; mode++)
    {
      std::vector<double> seen, done;
      for (int rep = 0; rep < 60; rep++)
      {
        tag++;
        const long n_per_wg = (long)mb * (1l << 20) / 4 / wgs;
        const double t0 = now_us();
        if (mode >= 2 && mode < 6)
          hipExtLaunchKernelGGL(k, dim3(wgs), dim3(1024), 0, s, nullptr, ev, 0, buf, n_per_wg, d_flag, tag, 10000ll);
        else
          hipLaunchKernelGGL(k, dim3(wgs), dim3(1024), 0, s, buf, n_per_wg, d_flag, tag, 10000ll);
        if (mode < 6) // (modes 6, 7: synchronize at once, the marker queues up behind the running kernel)
          for (;;)
          {
            bool all = true;
            for (int w = 0; w < wgs && all; w++)
              all = __atomic_load_n(&h_flag[w], __ATOMIC_ACQUIRE) == tag;
            if (all)
              break;
          }
        const double t1 = now_us();
        if (mode == 0)
          hipDeviceSynchronize();
        else if (mode == 1)
          hipStreamSynchronize(s);
        else if (mode == 2)
          hipEventSynchronize(ev);
        else if (mode == 3)
          while (hipStreamQuery(s) == hipErrorNotReady) {}
        else if (mode == 4)
        {
          hipEventSynchronize(ev);
          hipDeviceSynchronize(); // what a host that fences per burst does behind nam_hip_batch_flush
        }
        else if (mode == 5)
        {
          hipEventSynchronize(ev);
          hipStreamSynchronize(s);
        }
        else if (mode == 6)
          hipDeviceSynchronize();
        else
          hipStreamSynchronize(s);
        const double t2 = now_us();
        if (rep >= 10)
        {
          seen.push_back(t1 - t0);
          done.push_back(t2 - t1);
        }
      }
      std::sort(seen.begin(), seen.end());
      std::sort(done.begin(), done.end());
      const char* names[8] = {"plain + hipDeviceSynchronize", "plain + hipStreamSynchronize", "ext stop event + hipEventSynchronize", "ext stop event + hipStreamQuery spin",
                              "ext stop event + EventSync + DeviceSync", "ext stop event + EventSync + StreamSync",
                              "DeviceSync AT ONCE (no polling first)", "StreamSync AT ONCE (no polling first)"};
      std::vector<double> tot(seen.size());
      for (size_t i = 0; i < seen.size(); i++)
        tot[i] = seen[i] + done[i];
      std::sort(tot.begin(), tot.end());
      printf("%2d MB dirtied, %-40s launch -> flags visible %7.2f us (median), flags visible -> synchronized %6.2f us (median; min %5.2f, p90 %5.2f); launch -> synchronized %7.2f\n", mb, names[mode],
             seen[seen.size() / 2], done[done.size() / 2], done[0], done[done.size() * 9 / 10], tot[tot.size() / 2]);
    }
  return 0;
}
