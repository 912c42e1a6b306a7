// developer probe: vector-ALU issue cost on gfx950 — v_fma_f32 vs v_pk_fma_f32 vs v_rcp_f32 (independent chains), alone and
// next to fp32 MFMAs, with 1 / 2 / 3 / 4 waves per SIMD (waves i, i + 4, i + 8, i + 12 of a workgroup share a SIMD), and the
// instruction mix of nam_a1_q_kernel's stage bodies (modes 7, 8): what an "other vector instruction" costs the issue port NEXT TO
// the kernel's own matrix instructions at the kernel's own occupancy — bench.py's issue floor uses that figure
// (profiles/traffic.json: issue_cycles_per_valu_inst).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rate tools/src/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// MODE 0: 8 independent v_fma_f32 per iteration; 1: 8 v_pk_fma_f32 (16 FMAs); 2: 8 v_rcp_f32; 3: 4 MFMA 16x16x4 + 8 v_fma;
// 4: 4 MFMA 16x16x4 alone; 5: 8 v_fma + 8 v_mul interleaved (16 plain); 6: 4 MFMA 4x4x1 + 8 v_fma;
// 7: a big-stage job of nam_a1_q_kernel (one layer on a 16-frame sub-block): 16 MFMA 16x16x4 + 4 x (9 plain + 1 v_rcp) activation
//    + 12 plain (bias / head / residual);  8: a small-stage job (one 8-channel layer, 64 frames): 64 MFMA 4x4x1 + 8 x (9 + 1 rcp) + 16 plain;
// 9: 64 MFMA 4x4x1 alone
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* p, long long* cyc, int iters)
{
  float a[8], m = p[threadIdx.x & 63], c = p[64 + (threadIdx.x & 63)];
  f2 q[8];
  f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
  for (int i = 0; i < 8; i++)
  {
    a[i] = p[128 + i + (threadIdx.x & 63)];
    q[i] = f2{a[i], a[i] + 1.0f};
  }
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; it++)
  {
    if (MODE == 0 || MODE == 3 || MODE == 6)
    {
#pragma unroll
      for (int i = 0; i < 8; i++)
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
    }
    if (MODE == 5)
    {
#pragma unroll
      for (int i = 0; i < 8; i++)
      {
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(q[i][0]) : "v"(m));
      }
    }
    if (MODE == 1)
    {
#pragma unroll
      for (int i = 0; i < 8; i++)
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(q[i]) : "v"(f2{m, m}), "v"(f2{c, c}));
    }
    if (MODE == 2)
    {
#pragma unroll
      for (int i = 0; i < 8; i++)
        asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
    }
    if (MODE == 3 || MODE == 4)
    {
#pragma unroll
      for (int i = 0; i < 4; i++)
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(m, c, acc[i], 0, 0, 0);
    }
    if (MODE == 6)
    {
#pragma unroll
      for (int i = 0; i < 4; i++)
        acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(m, c, acc[i], 0, 0, 0);
    }
    if (MODE == 7)
    {
#pragma unroll
      for (int t = 0; t < 3; t++)
#pragma unroll
        for (int i = 0; i < 4; i++)
          acc[t & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(m, a[i], acc[t & 1], 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 4; e++)
      {
#pragma unroll
        for (int i = 0; i < 9; i++)
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[(e + i) & 7]) : "v"(m), "v"(c));
        asm volatile("v_rcp_f32 %0, %0" : "+v"(a[e]));
      }
#pragma unroll
      for (int i = 0; i < 12; i++)
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i & 7]) : "v"(m), "v"(c));
#pragma unroll
      for (int i = 0; i < 4; i++)
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(m, a[i], acc[2], 0, 0, 0);
    }
    if (MODE == 8 || MODE == 9)
    {
#pragma unroll
      for (int t = 0; t < 3; t++)
#pragma unroll
        for (int i = 0; i < 16; i++)
          acc[i & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(m, a[i & 7], acc[i & 1], 0, 0, 0);
      if (MODE == 8)
      {
#pragma unroll
        for (int e = 0; e < 8; e++)
        {
#pragma unroll
          for (int i = 0; i < 9; i++)
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[(e + i) & 7]) : "v"(m), "v"(c));
          asm volatile("v_rcp_f32 %0, %0" : "+v"(a[e]));
        }
#pragma unroll
        for (int i = 0; i < 16; i++)
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i & 7]) : "v"(m), "v"(c));
      }
#pragma unroll
      for (int i = 0; i < 16; i++)
        acc[2 + (i & 1)] = __builtin_amdgcn_mfma_f32_4x4x1f32(m, a[i & 7], acc[2 + (i & 1)], 0, 0, 0);
    }
  }
  const long long t1 = clock64();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++)
    s += a[i] + q[i][0] + q[i][1];
  p[threadIdx.x & 63] = s + acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
  // every wave's own start and end: the SIMD's time is the span over ITS waves (the arbiter favours the oldest wave: wave 0's
  // own time undercounts — round 4's three-wave rows did, e.g. 21 cycles per fp32 16x16x4 MFMA, which takes 32)
  if ((threadIdx.x & 63) == 0)
  {
    cyc[2 * (threadIdx.x >> 6)] = t0;
    cyc[2 * (threadIdx.x >> 6) + 1] = t1;
  }
}
template <int MODE>
void run(const char* name, float* d, long long* dc, int waves_per_simd, double per_iter)
{
  const int iters = 512;
  for (int rep = 0; rep < 2; rep++)
    hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(256 * waves_per_simd), 0, 0, d, dc, iters);
  long long cs[32];
  hipMemcpy(cs, dc, sizeof(cs), hipMemcpyDeviceToHost);
  // SIMD 0 holds waves 0, 4, 8, 12 (tools/src/simd_map.hip): its span
  long long lo = cs[0], hi = cs[1];
  for (int w = 0; w < 4 * waves_per_simd; w += 4)
  {
    lo = cs[2 * w] < lo ? cs[2 * w] : lo;
    hi = cs[2 * w + 1] > hi ? cs[2 * w + 1] : hi;
  }
  const long long c = hi - lo;
  printf("%-46s %d wave(s)/SIMD: %7.2f cycles per iteration per SIMD (%5.2f per instruction of one wave, %5.2f per SIMD-instruction)\n", name,
         waves_per_simd, (double)c / iters, (double)c / iters / per_iter, (double)c / iters / per_iter / waves_per_simd);
}
int main()
{
  float* d;
  long long* dc;
  hipMalloc(&d, 4096);
  hipMalloc(&dc, 32 * 8);
  hipMemset(d, 0, 4096);
  for (int w = 1; w <= 4; w++)
  {
    run<0>("8 x v_fma_f32", d, dc, w, 8);
    run<5>("8 x (v_fma_f32 + v_mul_f32)", d, dc, w, 16);
    run<1>("8 x v_pk_fma_f32", d, dc, w, 8);
    run<2>("8 x v_rcp_f32", d, dc, w, 8);
    run<4>("4 x mfma 16x16x4", d, dc, w, 4);
    run<3>("4 x mfma 16x16x4 + 8 x v_fma_f32", d, dc, w, 12);
    run<6>("4 x mfma 4x4x1 + 8 x v_fma_f32", d, dc, w, 12);
    run<7>("a1_q big job: 16 mfma 16x16x4 + 52 valu", d, dc, w, 68);
    run<9>("64 x mfma 4x4x1", d, dc, w, 64);
    run<8>("a1_q small job: 64 mfma 4x4x1 + 96 valu", d, dc, w, 160);
  }
  // the figure bench.py's issue floor wants: cycles of the port per non-matrix vector instruction NEXT TO the matrix instructions,
  // at four waves per SIMD = (cycles per iteration per SIMD - the matrix instructions' own cycles) / the other instructions
  printf("issue floor: see the `4 wave(s)/SIMD` rows of the two a1_q jobs: (cycles per iteration - 4 x 16 x 32) / (4 x 52) for the big job, "
         "(cycles - the `64 x mfma 4x4x1` row x 80 / 64) / (4 x 96) for the small one\n");
  return 0;
}
