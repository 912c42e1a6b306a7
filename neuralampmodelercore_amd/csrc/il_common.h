// il_common.h — helpers shared by the interleaved-frame kernels (kernel_a1_p2.hip and the pipelines built on its lane layout and
// session protocol: kernel_a1_p4.hip, kernel_a1_q.hip, kernel_kq.hip). The descriptor-driven form of the mapping,
// nam_a1_il_kernel, was retired in round 5: topologies outside the compile-time tables run nam_a1_mfma_kernel.
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

// A session launch waits for command `tag` (ring slot of sequence number tag - 1; `v` = what the slot held at the last look) for
// `window` ticks of the 100 MHz clock. A launch that LINGERS (ticketed host buffers: A1Args::p_linger > 100; nam_a1_q_kernel,
// nam_kq_kernel) has two more rules:
//  * the host's "leave" word behind the ring (= the session's command count: nothing follows) ends the wait at once;
//  * when the window has passed, the workgroup still stays while another workgroup is BEHIND it — p_cmd_count[mask + 1] holds
//    the highest command every workgroup is through (atomic max by the last one through each). The host hands the next buffer
//    in when the slowest workgroup has finished an old one; a workgroup that left meanwhile would miss it, the rest would have
//    to run dry, linger and leave before the next launch could pick it up again — and then the two halves take turns being the
//    one that left (measured: 1,024-frame buffers at 27 k xRT instead of 57 k, a relaunch per 16 buffers). Not once a workgroup
//    of this launch HAS left (p_cmd_count[mask + 2], cleared by the host in front of every launch: session_leaving): the one
//    behind may be the one that is gone. Capped at 20 ms.
template <int SLEEP, class RingLoad>
__device__ __forceinline__ unsigned long long session_wait_command(const A1Args& a, RingLoad ring_load, const unsigned tag, unsigned long long v,
                                                                  const long long window)
{
  if ((unsigned)(v >> 32) == tag || window <= 0)
    return v;
  const bool lingers = a.p_linger > 100;
  const long long t0 = (long long)wall_clock64(), t_end = t0 + window, t_cap = t0 + (lingers ? 2000000ll : window);
  int why = 0; // (developer statistics: why the wait ended without a command — NAM_HIP_SESSION_STATS, a.dbg = a host-mapped array)
  for (;;)
  {
    __builtin_amdgcn_s_sleep(SLEEP);
    v = ring_load(tag - 1u);
    if ((unsigned)(v >> 32) == tag)
      break;
    const long long now = (long long)wall_clock64();
    if (!lingers)
    {
      if (now >= t_end)
        break;
      continue;
    }
    if ((unsigned)__hip_atomic_load(a.p_ring + a.p_ring_mask + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == tag - 1u)
    {
      why = 1;
      break;
    }
    if (now >= t_end)
    {
      const unsigned all_done = __hip_atomic_load(a.p_cmd_count + a.p_ring_mask + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      why = now >= t_cap ? 2 : (int)(all_done - (tag - 1u)) >= 0 ? 3 : __hip_atomic_load(a.p_cmd_count + a.p_ring_mask + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ? 4 : 0;
      if (why)
      {
        if (a.dbg && a.p_ring)
          a.dbg[blockIdx.x] = ((long long)why << 56) | ((long long)(SLEEP == 8 ? 1 : 0) << 48) | ((long long)(all_done & 0xffffffu) << 24) | (long long)((tag - 1u) & 0xffffffu);
        break;
      }
    }
  }
  (void)why;
  return v;
}
// ... and a workgroup of a lingering launch that leaves says so (one lane)
__device__ __forceinline__ void session_leaving(const A1Args& a)
{
  if (a.p_linger > 100)
    __hip_atomic_fetch_add(a.p_cmd_count + a.p_ring_mask + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
} // namespace namhip
