// developer probe: which SIMD does wave i of a workgroup run on? (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8], ...)
// nam_kq_kernel / nam_a1_q_kernel place their stages on the assumption that waves i, i + 4, i + 8 share a SIMD.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/simd_map tools/src/simd_map.hip && /tmp/simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out)
{
  extern __shared__ float lds[];
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0)
    out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
  lds[threadIdx.x] = 0;
}
int main()
{
  unsigned* d;
  hipMalloc(&d, 64 * 16 * 4);
  for (int waves : {12, 16, 8})
    for (int lds : {0, 150 * 1024})
    {
      hipMemset(d, 0xff, 64 * 16 * 4);
      hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL(k, dim3(8), dim3(waves * 64), lds + 4096, 0, d);
      unsigned h[64 * 16];
      hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      printf("%d waves per workgroup, %d KB of LDS: SIMD of wave 0, 1, ... (8 workgroups)\n", waves, (lds + 4096) / 1024);
      for (int b = 0; b < 8; b++)
      {
        printf("   wg %d (cu %2u):", b, (h[b * 16] >> 8) & 15);
        for (int w = 0; w < waves; w++)
          printf(" %u", (h[b * 16 + w] >> 4) & 3);
        printf("\n");
      }
    }
  return 0;
}
