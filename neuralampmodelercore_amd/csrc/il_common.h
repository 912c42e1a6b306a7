// il_common.h — helpers shared by the interleaved-frame kernels (kernel_a1_il.hip: descriptor-driven,
// kernel_a1_p2.hip: compile-time topology).
#pragma once
#include "device_common.h"

#include <utility>

namespace namhip
{
namespace il
{
using mf::f2;
using mf::f4;
using u4 = __attribute__((ext_vector_type(4))) unsigned;
struct Slot
{
  f4 a, b; // raw ring rows (16 bytes of the lane's channel quad) of the job's two requests
  float inp; // input sample of the lane's frame in the block the job belongs to (every job carries it: a load inside
             // an `if (block boundary)` would turn every counted vmcnt wait behind it into vmcnt(0))
};
struct Ops // one job's register-resident operands
{
  f4 t[4]; // A tiles: conv tap 0, 1, 2 | layer1x1
  f4 xt; // extra tile (rechannel / head rechannel)
  f4 bv, mv, b1v, ev; // conv bias | input mixin | 1x1 bias | extra
  int ready; // loader progress word as read just before the operands
};
constexpr unsigned kOob = 0x7ffffff0u; // beyond num_records: loads return 0, stores are dropped, no memory traffic

template <int N>
__device__ __forceinline__ float row_shr(float old, float src)
{
  // lane i of a 16-lane row receives src of lane i - N; lanes i < N keep `old`
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src),
                                                               0x110 + N, 0xf, 0xf, false));
}
// taps of a layer with dilation 4 * N1 from the lane's own registers: tap 1 = frame t - d (N1 lanes down the row),
// tap 0 = frame t - 2d (2 N1 lanes down; a whole row = the previous block's same lane). `pa` / `pb`: the previous
// block's values for the lanes that fall off the row. NV values per lane (4: full layout, 2: half layout).
template <int NV, int N1>
__device__ __forceinline__ void dpp_taps(const f4& x, const f4& pa, const f4& pb, f4& bt0, f4& bt1)
{
#pragma unroll
  for (int e = 0; e < NV; e++)
  {
    bt1[e] = row_shr<N1>(pb[e], x[e]);
    if constexpr (2 * N1 < 16)
      bt0[e] = row_shr<2 * N1>(pa[e], x[e]);
    else
      bt0[e] = pa[e];
  }
}
// f(std::integral_constant<int, 0>{}), f(<1>), ... f(<N-1>) in order: a compile-time unrolled job sequence
template <class F, int... I>
__device__ __forceinline__ void for_each_index(F&& f, std::integer_sequence<int, I...>)
{
  (f(std::integral_constant<int, I>{}), ...);
}
} // namespace il
} // namespace namhip
