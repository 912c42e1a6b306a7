// Test helper (tests/test_gpu_tickets.py: test_watchdog_*): keeps every CU's LDS for `ms` milliseconds, so that another
// process's resident launch (144 KB of LDS per workgroup) cannot be placed — the "launch makes no progress" case of the
// session watchdog (NAM_HIP_PERSIST_TIMEOUT_MS). Prints "running" once the first workgroup executes, "done" when it is over.
//   hipcc --offload-arch=gfx950 -O2 -o gpu_hog gpu_hog.hip && ./gpu_hog 400
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(1024) void hog(unsigned* flag, long long ticks)
{
  extern __shared__ float keep[];
  keep[threadIdx.x] = 1.0f;
  if (threadIdx.x == 0)
    __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const long long t0 = wall_clock64(); // 100 MHz
  while ((long long)wall_clock64() - t0 < ticks)
    __builtin_amdgcn_s_sleep(64);
  if (keep[threadIdx.x] < 0.0f)
    flag[1] = 2u;
}
int main(int argc, char** argv)
{
  const long ms = argc > 1 ? std::atol(argv[1]) : 300;
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  unsigned* h_flag = nullptr;
  (void)hipHostMalloc((void**)&h_flag, 64, hipHostMallocMapped | hipHostMallocCoherent);
  h_flag[0] = h_flag[1] = 0;
  unsigned* d_flag = nullptr;
  (void)hipHostGetDevicePointer((void**)&d_flag, h_flag, 0);
  const int lds = 160 * 1024;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&hog), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(hog, dim3(cus), dim3(1024), lds, 0, d_flag, (long long)ms * 100000ll);
  if (hipGetLastError() != hipSuccess)
    return 2;
  while (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) == 0u) {}
  std::printf("running\n");
  std::fflush(stdout);
  const hipError_t e = hipDeviceSynchronize();
  std::printf("done %d\n", (int)e);
  return e == hipSuccess ? 0 : 3;
}
