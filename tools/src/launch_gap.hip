// Developer probe: cadence of back-to-back dependent kernel launches on one stream (empty kernel and a kernel that
// spins for a given number of microseconds), i.e. how much of a short kernel's launch-to-launch time is dispatch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void spin_kernel(long long cycles)
{
  const long long t0 = wall_clock64();
  while ((long long)wall_clock64() - t0 < cycles) {}
}
int main(int argc, char** argv)
{
  const int n = 2000;
  hipStream_t s;
  hipStreamCreate(&s);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (long long us : {0LL, 5LL, 10LL, 15LL})
    for (int blocks : {1, 256})
    {
      const long long cyc = us * 100; // wall_clock64 ticks at 100 MHz
      for (int i = 0; i < 200; i++)
        hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(512), 0, s, cyc);
      hipStreamSynchronize(s);
      hipEventRecord(e0, s);
      for (int i = 0; i < n; i++)
        hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(512), 0, s, cyc);
      hipEventRecord(e1, s);
      hipStreamSynchronize(s);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      printf("spin %lld us, %d blocks: %.2f us per launch\n", us, blocks, ms * 1e3 / n);
    }
  return 0;
}
