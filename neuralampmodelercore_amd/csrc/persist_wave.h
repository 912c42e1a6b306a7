// persist_wave.h — device side of the persistent block mode for kernels whose workgroup is ONE wavefront
// (nam_wn_reg_kernel, nam_lstm_row_kernel). The protocol is the one nam_a1_p2_kernel speaks (kernel_a1_p2.hip, PERSIST;
// host side: api_session.cpp; api_internal.h: PersistSession): the launch consumes COMMANDS — one per 64-frame buffer, `(seq << 32) |
// frame offset` in a ring the host (or the caller's stream) stores into — for as long as the next one is already there
// when a buffer is finished, and leaves as soon as the ring is empty (it never waits unboundedly: a device-wide
// synchronize must not depend on a command arriving). A wavefront of its own needs no agreement step: every lane
// reads the same ring word.
#pragma once

#include <hip/hip_runtime.h>

#include "kernels.h"

namespace namhip
{

struct PersistWave
{
  unsigned seq = 0; // commands consumed so far
  unsigned long long spec = 0; // the early look at the next command

  static __device__ __forceinline__ unsigned long long ring_load(const PersistArgs& p, unsigned s)
  {
    const unsigned long long v = __hip_atomic_load(p.ring + (s & (unsigned)p.ring_mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // wavefront-uniform (every lane loaded the same word): keep it in scalar registers
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
  }
  // the first command of this launch; false = nothing to do (its predecessor consumed the command it was started for)
  __device__ __forceinline__ bool begin(const PersistArgs& p, int wg, unsigned& frame_off)
  {
    if (p.seq0 >= 0) // every workgroup stands at the same count: it and the first command came with the launch
    {
      seq = (unsigned)p.seq0;
      frame_off = (unsigned)p.cmd0;
      return true;
    }
    seq = (unsigned)__builtin_amdgcn_readfirstlane((int)p.cons[wg]);
    unsigned long long v = ring_load(p, seq);
    if (p.grace > 0 && (unsigned)(v >> 32) != seq + 1)
    {
      // started right behind a stream-ordered command store on another hardware queue: look for a bounded time
      const long long t_end = (long long)wall_clock64() + p.grace;
      do
      {
        __builtin_amdgcn_s_sleep(8);
        v = ring_load(p, seq);
      } while ((unsigned)(v >> 32) != seq + 1 && (long long)wall_clock64() < t_end);
    }
    frame_off = (unsigned)v;
    return (unsigned)(v >> 32) == seq + 1;
  }
  // while a buffer is being processed: request the next command (consumed by next())
  __device__ __forceinline__ void look_ahead(const PersistArgs& p) { spec = ring_load(p, seq + 1); }
  // a buffer is finished: the next command, or false when the ring is empty (the caller then leaves)
  __device__ __forceinline__ bool next(const PersistArgs& p, int wg, unsigned& frame_off)
  {
    seq++;
    if ((seq & 15u) == 0u && threadIdx.x == 0) // progress for the host's ring bookkeeping (no fence: not a completion signal)
      __hip_atomic_store(p.prog + wg, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    unsigned long long v = spec;
    if ((unsigned)(v >> 32) != seq + 1)
      v = ring_load(p, seq); // the early look missed: once more, now
    frame_off = (unsigned)v;
    return (unsigned)(v >> 32) == seq + 1;
  }
  // results visible, then the consumed-command count: device copy for this workgroup's next launch, host copy (with
  // the "left" bit) for the host
  __device__ __forceinline__ void leave(const PersistArgs& p, int wg)
  {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if (threadIdx.x == 0)
    {
      p.cons[wg] = seq;
      __hip_atomic_store(p.done + wg, seq | 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
};

// an input sample of a persistent session: past the caches (the caller may rewrite the same buffer between two commands
// and no kernel boundary invalidates what this CU read a buffer ago)
__device__ __forceinline__ float persist_in(const float* p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

} // namespace namhip
