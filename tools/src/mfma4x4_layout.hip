// developer probe: operand / result layout of v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4 x 4, K = 1)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma4x4 tools/src/mfma4x4_layout.hip && /tmp/mfma4x4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, float* d)
{
  f4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
  for (int e = 0; e < 4; e++)
    d[e * 64 + threadIdx.x] = acc[e];
}
int main()
{
  float ha[64], hb[64], hd[256];
  // A: lane l carries 1000 + l ; B: lane l carries l + 1 -> D = A_lane_x * B_lane_y tells which lanes met
  for (int l = 0; l < 64; l++)
  {
    ha[l] = (float)(100 + l);
    hb[l] = (float)(l + 1);
  }
  float *da, *db, *dd;
  hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
  hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
  hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; l++)
    for (int e = 0; e < 4; e++)
    {
      // expectation: D[e] of lane l = A of lane (l & ~3) + e  times  B of lane l
      const float want = (float)(100 + (l & ~3) + e) * (float)(l + 1);
      if (hd[e * 64 + l] != want)
        bad++;
    }
  printf("lane 5: D = %g %g %g %g (A lanes 4..7 = 104..107 times B lane 5 = 6 expected)\n", hd[5], hd[64 + 5], hd[128 + 5], hd[192 + 5]);
  printf("layout %s\n", bad ? "DIFFERS" : "as assumed: block = lane / 4, A row i and B column j = lane % 4, D[e] = row e");
  return bad != 0;
}
