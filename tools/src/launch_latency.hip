// Developer probe (round 6): host clock from the launch call to (a) the call's return, (b) the first wave of the grid writing a
// flag the host sees (host-mapped memory, system-scope store), (c) the last workgroup's flag — for the session launch's own shape
// (256 workgroups x 1,024 threads, 144 KB of LDS) and smaller ones, through hipLaunchKernelGGL, hipExtLaunchKernelGGL with a stop
// event (what the sessions use: kernels.h nam_launch) and with a big by-value argument like A1Args. The queue is idle at every
// launch (the sessions' case: a launch per burst).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/launch_latency tools/src/launch_latency.hip && /tmp/launch_latency
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
struct Big { long long pad[40]; unsigned* first; unsigned* all; unsigned* cnt; unsigned tag; };
__global__ __launch_bounds__(1024) void k(const Big a)
{
  extern __shared__ float lds[];
  if (threadIdx.x == 0)
  {
    if (blockIdx.x == 0)
      __hip_atomic_store(a.first, a.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    lds[0] = 1.0f;
    const unsigned before = __hip_atomic_fetch_add(a.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (before == gridDim.x - 1u)
    {
      __hip_atomic_store(a.cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.all, a.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
  hipStream_t s;
  int lo, hi;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  (void)hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi);
  unsigned* h = nullptr;
  (void)hipHostMalloc((void**)&h, 64, hipHostMallocMapped | hipHostMallocCoherent);
  unsigned* d = nullptr;
  (void)hipHostGetDevicePointer((void**)&d, h, 0);
  unsigned* dcount = nullptr;
  (void)hipMalloc((void**)&dcount, 64);
  (void)hipMemset(dcount, 0, 64);
  hipEvent_t stop;
  (void)hipEventCreateWithFlags(&stop, hipEventDisableTiming);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  struct Shape { int wgs, threads, lds; const char* name; } shapes[] = {
    {1, 64, 0, "1 x 64, no LDS"}, {256, 64, 0, "256 x 64, no LDS"}, {256, 1024, 0, "256 x 1024, no LDS"}, {256, 1024, 144 * 1024, "256 x 1024, 144 KB LDS (session)"}};
  for (const auto& sh : shapes)
    for (int how = 0; how < 2; how++)
    {
      std::vector<double> t_ret, t_first, t_all, t_retire;
      for (int rep = 0; rep < 300; rep++)
      {
        Big a{};
        a.first = d; a.all = d + 8; a.cnt = dcount; a.tag = (unsigned)rep + 1u;
        h[0] = h[8] = 0;
        (void)hipStreamSynchronize(s);
        const double t0 = now_us();
        if (how == 0)
          hipLaunchKernelGGL(k, dim3(sh.wgs), dim3(sh.threads), sh.lds, s, a);
        else
          hipExtLaunchKernelGGL(k, dim3(sh.wgs), dim3(sh.threads), sh.lds, s, nullptr, stop, 0, a);
        const double t1 = now_us();
        while (__atomic_load_n(&h[0], __ATOMIC_ACQUIRE) != a.tag) {}
        const double t2 = now_us();
        while (__atomic_load_n(&h[8], __ATOMIC_ACQUIRE) != a.tag) {}
        const double t3 = now_us();
        if (how == 1) (void)hipEventSynchronize(stop); else (void)hipStreamSynchronize(s);
        const double t4 = now_us();
        if (rep >= 20) { t_ret.push_back(t1 - t0); t_first.push_back(t2 - t0); t_all.push_back(t3 - t0); t_retire.push_back(t4 - t0); }
      }
      auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
      std::printf("%-36s %-24s call returns %6.2f us | first wave's flag %6.2f | last workgroup's flag %6.2f | retired %6.2f\n", sh.name,
                  how == 0 ? "hipLaunchKernelGGL" : "hipExtLaunchKernelGGL+event", med(t_ret), med(t_first), med(t_all), med(t_retire));
    }
  return 0;
}
