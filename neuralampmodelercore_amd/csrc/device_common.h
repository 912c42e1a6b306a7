// device_common.h — device-side helpers shared by the kernel translation units: the reference's activation
// formulas, wave-uniform broadcast, and the lane-layout / LDS / MFMA helpers of the matrix-core kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "kernels.h"

namespace namhip
{

// ------------------------------------------------------------------------------------------------
// Activations (NAM/activations.h:59-133)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float d_fast_tanh(const float x)
{
  const float ax = fabsf(x);
  const float x2 = x * x;
  return (x * (2.45550750702956f + 2.45550750702956f * ax + (0.893229853513558f + 0.821226666969744f * ax) * x2)
          / (2.44506634652299f + (2.44506634652299f + x2) * fabsf(x + 0.814642734961073f * x * ax)));
}
__device__ __forceinline__ float d_fast_sigmoid(const float x)
{
  return 0.5f * (d_fast_tanh(x * 0.5f) + 1.0f);
}
__device__ __forceinline__ float d_sigmoid(const float x)
{
  return 1.0f / (1.0f + expf(-x));
}

template <int TYPE>
__device__ __forceinline__ float d_act(float x, float p0, float p1, float p2, float p3, float slope)
{
  if constexpr (TYPE == ACT_TANH)
    return tanhf(x);
  else if constexpr (TYPE == ACT_HARDTANH)
  {
    const float t = x < -1.0f ? -1.0f : x;
    return t > 1.0f ? 1.0f : t;
  }
  else if constexpr (TYPE == ACT_FASTTANH)
    return d_fast_tanh(x);
  else if constexpr (TYPE == ACT_RELU)
    return x > 0.0f ? x : 0.0f;
  else if constexpr (TYPE == ACT_LEAKYRELU)
    return x > 0.0f ? x : p0 * x;
  else if constexpr (TYPE == ACT_PRELU)
    return x > 0.0f ? x : slope * x;
  else if constexpr (TYPE == ACT_SIGMOID)
    return d_sigmoid(x);
  else if constexpr (TYPE == ACT_SILU)
    return x * d_sigmoid(x);
  else if constexpr (TYPE == ACT_HARDSWISH)
  {
    const float t = x + 3.0f;
    const float c = t < 0.0f ? 0.0f : (t > 6.0f ? 6.0f : t);
    return x * c * (1.0f / 6.0f);
  }
  else if constexpr (TYPE == ACT_LEAKYHARDTANH)
  {
    if (x < p0)
      return (x - p0) * p2 + p0;
    else if (x > p1)
      return (x - p1) * p3 + p1;
    return x;
  }
  else if constexpr (TYPE == ACT_SOFTSIGN)
    return x / (1.0f + fabsf(x));
  else if constexpr (TYPE == ACT_FASTSIGMOID)
    return d_fast_sigmoid(x);
  else
    return x;
}

// FastLUTActivation::lookup (NAM/activations.h:391-409): clamp, index, linear interpolation. `tbl` points at the
// activation's parameter block in the blob: min_x, max_x, inv_step, n_points, then the table itself.
__device__ __forceinline__ float d_lut(const float* __restrict__ tbl, float x)
{
  const float min_x = tbl[0], max_x = tbl[1], inv_step = tbl[2];
  const unsigned n = (unsigned)tbl[3];
  x = x < min_x ? min_x : (x > max_x ? max_x : x);
  const float f_idx = (x - min_x) * inv_step;
  const unsigned i = (unsigned)f_idx;
  if (i >= n - 1u)
    return tbl[4 + n - 1u];
  const float frac = f_idx - (float)i;
  const float y0 = tbl[4 + i], y1 = tbl[5 + i];
  return y0 + (y1 - y0) * frac;
}

// run-time (wave-uniform) dispatch. An activation type a kernel does not know must never fall through to identity:
// it traps (the host-side eligibility checks in plan_a1.cpp are the only other guard).
__device__ __forceinline__ float d_act_rt(int type, float x, float p0, float p1, float p2, float p3, float slope)
{
  switch (type)
  {
    case ACT_TANH: return d_act<ACT_TANH>(x, p0, p1, p2, p3, slope);
    case ACT_HARDTANH: return d_act<ACT_HARDTANH>(x, p0, p1, p2, p3, slope);
    case ACT_FASTTANH: return d_act<ACT_FASTTANH>(x, p0, p1, p2, p3, slope);
    case ACT_RELU: return d_act<ACT_RELU>(x, p0, p1, p2, p3, slope);
    case ACT_LEAKYRELU: return d_act<ACT_LEAKYRELU>(x, p0, p1, p2, p3, slope);
    case ACT_PRELU: return d_act<ACT_PRELU>(x, p0, p1, p2, p3, slope);
    case ACT_SIGMOID: return d_act<ACT_SIGMOID>(x, p0, p1, p2, p3, slope);
    case ACT_SILU: return d_act<ACT_SILU>(x, p0, p1, p2, p3, slope);
    case ACT_HARDSWISH: return d_act<ACT_HARDSWISH>(x, p0, p1, p2, p3, slope);
    case ACT_LEAKYHARDTANH: return d_act<ACT_LEAKYHARDTANH>(x, p0, p1, p2, p3, slope);
    case ACT_SOFTSIGN: return d_act<ACT_SOFTSIGN>(x, p0, p1, p2, p3, slope);
    case ACT_FASTSIGMOID: return d_act<ACT_FASTSIGMOID>(x, p0, p1, p2, p3, slope);
    case ACT_IDENTITY: return x;
    default: __builtin_trap(); return x;
  }
}

__device__ __forceinline__ int uni(int v)
{
  return __builtin_amdgcn_readfirstlane(v);
}

using mf_f4 = __attribute__((ext_vector_type(4))) float;

// ------------------------------------------------------------------------------------------------
// A1-family MFMA kernel — shared helpers (the kernel itself, nam_a1_mfma_kernel, is further down)
// ------------------------------------------------------------------------------------------------
// Every matrix product of the model is (C x Kdim) * (Kdim x 64 frames); compute wave w owns frames
// [16w, 16w+16) and issues v_mfma_f32_16x16x4_f32 (exact fp32: bitwise an ordered fmaf chain).
// Lane l = (g = l >> 4, j = l & 15) of wave w, FULL layout (plan_a1.cpp describes the HALF layout of
// 8-channel arrays):
//   D (4 VGPR)  out channels 4g + r, r = 0..3, of frame 16w + j          (residual x, head, z live here)
//   B operand   k-step s feeds row k = g with channel 4g + s of frame 16w + j — THE LANE'S OWN D VALUES,
//               so the current tap, the 1x1, the rechannel and the head need no data movement at all
//   A operand   tile value W[out = j][in = 4g + s] (plan_a1.cpp packs the tiles for exactly this mapping)
// Only the time-shifted taps leave the registers: each lane fetches its channels of frame
// (16w + j - L) with ONE LDS read from a frame-major window (lookback L <= 64: [previous 64 |
// current 64] frames) or tap buffer (L > 64), both filled from the stream's frame-major history ring
// in HBM with 16-byte accesses.
namespace mf
{
using f4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ float rcp(float x)
{
  return __builtin_amdgcn_rcpf(x);
}
// tanh(x) = 1 - 2 / (exp(2x) + 1) on the hardware exp2 / rcp units (abs error ~1e-7)
__device__ __forceinline__ float tanh_hw(float x)
{
  const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f); // exp(2x) = 2^(2x*log2(e))
  return 1.0f - 2.0f * rcp(e + 1.0f);
}
// The reference's fast_tanh (NAM/activations.h:29-41): x (a + a |x| + (b + c |x|) x^2) / (d + (d + x^2) |x + e x |x||), in ten
// instructions instead of eleven: |x + e x |x|| = |x| (1 + e |x|) (e > 0), and with q = x^2 + d the numerator / x is
// t1 q + (t2 - d t1), whose second factor is linear in |x| like t2 — x^2 never exists by itself.
__device__ __forceinline__ float fast_tanh_hw(const float x)
{
  constexpr float kA1 = (float)(2.45550750702956 - 2.44506634652299 * 0.821226666969744), kA0 = (float)(2.45550750702956 - 2.44506634652299 * 0.893229853513558);
  const float ax = fabsf(x);
  const float q = fmaf(ax, ax, 2.44506634652299f);
  const float w = fmaf(0.814642734961073f, ax, 1.0f);
  const float t1 = fmaf(0.821226666969744f, ax, 0.893229853513558f);
  const float t2 = fmaf(kA1, ax, kA0);
  const float den = fmaf(q, ax * w, 2.44506634652299f);
  return (fmaf(t1, q, t2) * rcp(den)) * x;
}
__device__ __forceinline__ float sigmoid_hw(float x)
{
  return rcp(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
// fast_sigmoid of the reference's LSTM (lstm.cpp:48-58: 0.5 (fast_tanh(x / 2) + 1)) on the hardware rcp
__device__ __forceinline__ float fast_sigmoid_hw(const float x)
{
  return 0.5f * (fast_tanh_hw(x * 0.5f) + 1.0f);
}
__device__ __forceinline__ float act_hw(int type, float x, float p0)
{
  switch (type)
  {
    case ACT_TANH: return tanh_hw(x);
    case ACT_FASTTANH: return fast_tanh_hw(x);
    case ACT_HARDTANH: return fminf(fmaxf(x, -1.0f), 1.0f);
    case ACT_RELU: return x > 0.0f ? x : 0.0f;
    case ACT_LEAKYRELU: return x > 0.0f ? x : p0 * x;
    case ACT_SIGMOID: return sigmoid_hw(x);
    case ACT_SILU: return x * sigmoid_hw(x);
    case ACT_HARDSWISH:
    {
      const float t = fminf(fmaxf(x + 3.0f, 0.0f), 6.0f);
      return x * t * (1.0f / 6.0f);
    }
    case ACT_SOFTSIGN: return x * rcp(1.0f + fabsf(x));
    case ACT_IDENTITY: return x;
    default: __builtin_trap(); return x; // unknown type: never a silent identity (the planner gates what reaches here)
  }
}
// whole-vector activation; ACT_T >= 0 resolves the type at compile time (the two kernels that matter:
// Fasttanh = benchmodel default, Tanh), ACT_T < 0 dispatches once per job on the run-time type
template <int ACT_T>
__device__ __forceinline__ f4 act4(int type, const f4& v, float p0)
{
  f4 r;
  if constexpr (ACT_T == ACT_FASTTANH)
  {
#pragma unroll
    for (int i = 0; i < 4; i++)
      r[i] = fast_tanh_hw(v[i]);
  }
  else if constexpr (ACT_T == ACT_TANH)
  {
#pragma unroll
    for (int i = 0; i < 4; i++)
      r[i] = tanh_hw(v[i]);
  }
  else if constexpr (ACT_T == ACT_LEAKYRELU)
  {
#pragma unroll
    for (int i = 0; i < 4; i++)
      r[i] = v[i] > 0.0f ? v[i] : p0 * v[i];
  }
  else
  {
    // one dispatch per vector, not per element (a 10-way compare cascade per element costs hundreds of cycles on a
    // lone wavefront)
#define NAM_ACT4_CASE(T) \
  case T: \
    _Pragma("unroll") for (int i = 0; i < 4; i++) r[i] = act_hw(T, v[i], p0); \
    break;
    switch (type)
    {
      NAM_ACT4_CASE(ACT_TANH)
      NAM_ACT4_CASE(ACT_FASTTANH)
      NAM_ACT4_CASE(ACT_HARDTANH)
      NAM_ACT4_CASE(ACT_RELU)
      NAM_ACT4_CASE(ACT_LEAKYRELU)
      NAM_ACT4_CASE(ACT_SIGMOID)
      NAM_ACT4_CASE(ACT_SILU)
      NAM_ACT4_CASE(ACT_HARDSWISH)
      NAM_ACT4_CASE(ACT_SOFTSIGN)
      case ACT_IDENTITY: r = v; break;
      default: __builtin_trap(); r = v; break;
    }
#undef NAM_ACT4_CASE
  }
  return r;
}
// workgroup barrier that orders LDS traffic only: outstanding global loads stay in flight
__device__ __forceinline__ void lds_barrier()
{
  // the builtin, not inline asm: hipcc's waitcnt pass then knows nothing is outstanding on lgkmcnt after the
  // barrier; with an asm wait it re-waits (lgkmcnt(0)) before the first use of any register loaded in the previous job,
  // i.e. in front of the own-tap MFMA chain that is there to cover the latency of the shifted-tap reads
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0); vmcnt / expcnt untouched
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
template <int NK>
__device__ __forceinline__ f4 mfma_n(const f4& a, const f4& b, f4 acc)
{
#pragma unroll
  for (int s = 0; s < NK; s++)
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc, 0, 0, 0);
  return acc;
}
using f2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ f4 lds_ld4(const char* lds, unsigned byte_off)
{
  return *reinterpret_cast<const f4*>(lds + byte_off);
}
__device__ __forceinline__ void lds_st4(char* lds, unsigned byte_off, const f4& v)
{
  *reinterpret_cast<f4*>(lds + byte_off) = v;
}
} // namespace mf

// Ring append: 16 B at (wave-uniform base + per-lane byte offset). WT = write-through (sc0 sc1): a launch that
// covers only a block or two would otherwise leave every ring line dirty in L2 and pay for the write-back
// when the kernel retires (measured 3 us of a 28 us launch at 256 streams); long launches keep the default
// write-back policy, which is ~3% faster in steady state.
template <bool WT>
__device__ __forceinline__ void ring_store(char* base, unsigned off, mf::f4 v)
{
  if constexpr (WT)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v),
                                           __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000),
                                           (int)off, 0, /*sc0 sc1*/ 17);
  else
    *reinterpret_cast<mf::f4*>(base + off) = v;
}

} // namespace namhip
