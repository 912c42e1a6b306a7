// developer probe: issue rate of fp32 MFMA shapes, dependent chain vs four independent chains, one wave per SIMD
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_rate tools/src/mfma_rate.hip && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int SHAPE, int CHAINS>
__global__ void k(float* p, long long* cyc)
{
  f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float a = p[threadIdx.x], b = p[64 + threadIdx.x];
  const long long t0 = clock64();
  for (int i = 0; i < 256; i++)
  {
#pragma unroll
    for (int u = 0; u < 8; u++)
    {
      const int c = CHAINS == 1 ? 0 : (u & 3);
      if (SHAPE == 0)
        acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 0, 0, 0);
      else
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
  }
  const long long t1 = clock64();
  p[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
  if (threadIdx.x == 0)
    cyc[0] = t1 - t0;
}
template <int SHAPE, int CHAINS>
void run(const char* name, float* d, long long* dc)
{
  hipLaunchKernelGGL((k<SHAPE, CHAINS>), dim3(1), dim3(64), 0, 0, d, dc);
  hipLaunchKernelGGL((k<SHAPE, CHAINS>), dim3(1), dim3(64), 0, 0, d, dc);
  long long c;
  hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
  printf("%-44s %8.2f shader cycles per instruction\n", name, (double)c / 2048.0);
}
int main()
{
  float* d;
  long long* dc;
  hipMalloc(&d, 1024);
  hipMalloc(&dc, 8);
  hipMemset(d, 0, 1024);
  run<0, 1>("4x4x1_16b, one dependent chain", d, dc);
  run<0, 4>("4x4x1_16b, four independent chains", d, dc);
  run<1, 1>("16x16x4, one dependent chain", d, dc);
  run<1, 4>("16x16x4, four independent chains", d, dc);
  return 0;
}
