// kernel_a1_p3.hip — nam_a1_p3_kernel: nam_a1_p2_kernel with TWO wavefronts per SIMD — the two layer arrays of the
// official A1 topology run as two wave sets, software-pipelined across consecutive 64-frame buffers.
#include "device_common.h"
#include "il_common.h"

namespace namhip
{

// ================================================================================================
// What nam_a1_p2_kernel (kernel_a1_p2.hip; read its header first) leaves on the table: a block is a chain of 20
// dependent jobs on ONE wavefront per SIMD. Counters of the headline shape (profiles/r02/rocprofv3_summary_c2_p2.txt):
// the matrix pipe is busy a third of the time, 38 % of the wave cycles are issue stalls, 31 % waits — latency nobody
// covers. A WaveNet block has no parallelism left inside a stream (frames are already spread over the four waves), but
// CONSECUTIVE buffers of a stream do: array 1 of buffer k only needs array 0 of buffer k, so while it runs, array 0 of
// buffer k + 1 can run beside it.
//
//   set A = waves 0-3: jobs 0-9  (array 0, C0 channels) of buffer r      in round r
//   set B = waves 4-7: jobs 10-19 (array 1, C1 channels) of buffer r - 1  in round r
//
// Wave w of either set owns frames t = 4 j + w with the same lane layout, so the hand-over after job 9 — the
// rechannelled layer output x (model.cpp:536-545 + the next array's rechannel, :488), the head accumulator of array 0
// (:513-531; job 10 turns it into array 1's start value, :476-484) and the input sample (the condition, :839) — is lane
// to lane through LDS: 36 bytes per lane, double buffered. The sets are decoupled except for three workgroup barriers
// per round: the two exchange jobs of each array (dilations 1 and 2: jobs 0 / 1 and 10 / 11 sit in the same position of
// their set, so the barriers pair up) and the round boundary (hand-over + the next command). Each set keeps request
// slots for its own ten jobs only (80 VGPRs instead of 160), which is what lets two of these waves share a SIMD's 512
// registers. A wave set without work in a round (B in the first round, A in the drain round) meets the barriers anyway.
// Same state (rings, write positions), weights, LDS tiles and numerics as nam_a1_p2_kernel — MFMA for MFMA the same
// sums — so the two alternate freely between launches of one stream.
//
// PERSIST: the command protocol of nam_a1_p2_kernel (one command per buffer, leave when the ring is empty, never wait)
// with the look-ahead moved into set A: every A wave looks at the next ring slot in job 4 and, on a hit, requests the
// next buffer's input sample itself in job 6 (a hit is always THE next command, so the speculation cannot be wrong);
// wave 0 decides at the end of its round (looking once more when job 4 missed; polling for a few microseconds while it
// idles in a drain round) and the round-boundary barrier makes its decision the workgroup's.
// ================================================================================================
namespace p3
{
constexpr int kWinB = 2 * kBlock * 80; // one exchange window: [previous 64 | current 64] frames, rows of 80 bytes (p2's order)
constexpr int kWindows = 4; // set A: 0, 1; set B: 2, 3
constexpr int kConstsB = kWindows * kWinB;
constexpr int kXtB = kConstsB + p2::kJobs * 256;
constexpr int kTilesB = kXtB + p2::kXt * 1024;
constexpr int kFlagB = kTilesB + p2::kJobs * kWsTileFloats * 4;
constexpr int kHandB = kFlagB + 64; // hand-over [2 buffers]: x [256 lanes] f4 | head [256] f4 | cond [256] float
constexpr int kHandBufB = 256 * 16 + 256 * 16 + 256 * 4;
constexpr int kLdsBytes = kHandB + 2 * kHandBufB;
static_assert(kLdsBytes <= 160 * 1024, "p3 LDS layout");
} // namespace p3

template <int C0, int C1, int ACT_T, bool WT, bool PERSIST>
__global__ __launch_bounds__(512) void nam_a1_p3_kernel(const float* __restrict__ blob, const A1Args a)
{
  using namespace mf;
  using il::kOob;
  using il::Ops;
  constexpr int NJ = p2::kJobs, NH = NJ / 2;
  extern __shared__ __attribute__((aligned(16))) float lds_p3[];
  char* const lds = reinterpret_cast<char*>(lds_p3);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w8 = uni(tid >> 6);
  const int S = w8 >> 2; // wave set: 0 = A (array 0), 1 = B (array 1)
  const int w = w8 & 3;
  const int tid_s = tid & 255; // thread index inside the set
  const int stream = a.stream_map ? a.stream_map[blockIdx.x] : (int)blockIdx.x;
  float* st = a.state + (size_t)stream * a.state_stride;
  const int n_blocks = PERSIST ? (1 << 30) : (a.n_frames + kBlock - 1) / kBlock;

  const int g = lane >> 4;
  const int j = lane & 15;
  const int t = 4 * j + w; // this lane's frame inside the block
  const float* in = a.in ? a.in + (size_t)stream * a.io_stride : nullptr;
  float* out = a.out ? a.out + (size_t)stream * a.io_stride : nullptr;
  const float head_scale = a.head_scale;
  const float act_p0 = a.act_p0;
  const unsigned v_g16 = (unsigned)g * 16u;
  const unsigned v_gh8 = (unsigned)(g & 1) * 16u + (unsigned)(g >> 1) * 8u;
  const unsigned v_lane16 = (unsigned)lane * 16u;
  const bool hi_pair = (g >> 1) != 0;
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)st, 0, (int)(a.state_stride * 4), 0x00020000);
  const int io_bytes = PERSIST ? 0x7ffffff0 : a.n_frames * 4;
  const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc((void*)(in ? in : st), 0, in ? io_bytes : 0, 0x00020000);
  const auto rsrc_out = __builtin_amdgcn_make_buffer_rsrc((void*)(out ? out : st), 0, out ? io_bytes : 0, 0x00020000);
  using i4 = __attribute__((ext_vector_type(4))) int;
  // rsrc_in once more, as plain dwords (base, stride 0, num_records, flags): for the one load issued from inline asm
  const unsigned long long in_addr = (unsigned long long)(in ? in : st);
  const i4 in_desc = {uni((int)(unsigned)in_addr), uni((int)(unsigned)(in_addr >> 32) & 0xffff), in ? io_bytes : 0, 0x00020000};
  int* wpos_tbl = reinterpret_cast<int*>(st);
  int wposv = lane < NJ ? wpos_tbl[lane] : 0; // lane r = write position of ring r (ring r = job r); a wave advances the
                                              // whole register by its own set's frames and only its set's lanes are used
  const int ring_len_v = 2 * (1 << (lane % p2::kLayers)) + kBlock;

  // ---- weights -> LDS, once per launch, by all eight waves (requested first, stored behind the ring requests) ----
  constexpr int kT4 = NJ * 256 / 512; // 16-byte tile records per thread
  f4 tl4[kT4];
  {
    const f4* __restrict__ tsrc = reinterpret_cast<const f4*>(blob + a.tiles_off);
#pragma unroll
    for (int i = 0; i < kT4; i++)
      tl4[i] = tsrc[i * 512 + tid];
  }
  const f4* __restrict__ csrc = reinterpret_cast<const f4*>(blob + a.consts_off);
  const f4* __restrict__ xsrc = reinterpret_cast<const f4*>(blob + a.xt_off);
  const f4 c0v = csrc[min(tid, NJ * 16 - 1)]; // 320 constant records
  const f4 x0v = xsrc[min(tid, p2::kXt * 64 - 1)]; // 192 extra-tile records

  // the ring requests of job TJ for the block AHEAD blocks after the one the write positions stand at
  auto fetch = [&](f4& ra, f4& rb, auto tj_tag, auto ahead_tag, bool valid, int tl, unsigned gl16) {
    constexpr int JF = (decltype(tj_tag)::value + NJ - p2::kDepth) % NJ; // table position whose entry describes job TJ
    constexpr int AHEAD = decltype(ahead_tag)::value;
    constexpr IlFetch F = p2::fetch(C0, C1, JF);
    int wp = __builtin_amdgcn_readlane(wposv, F.ring_id) + (AHEAD ? kBlock : 0);
    if (wp >= F.R)
      wp -= F.R;
    constexpr bool half = F.row_b == 32;
    const unsigned chan = min(half ? (gl16 & 16u) : gl16, (unsigned)F.row_b - 16u);
    const unsigned base = (unsigned)F.ring_b + chan;
#pragma unroll
    for (int q = 0; q < 2; q++)
    {
      constexpr int LA = F.LA, LB = F.LB, nA = F.nA, nB = F.nB;
      const int L = q == 0 ? LA : LB;
      const int n = q == 0 ? nA : nB;
      if (L > 0) // (compile time: exchange jobs have one request)
      {
        const unsigned v = (unsigned)(wp + tl - L + F.R);
        const unsigned idx = min(v, v - (unsigned)F.R);
        const bool want = valid && tl < 4 * n; // lanes j < n
        const unsigned off = want ? __umul24(idx, (unsigned)F.row_b) + base : kOob;
        const f4 r = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0));
        if (q == 0)
          ra = r;
        else
          rb = r;
      }
    }
  };
  // Request slots: one per job of the wave's OWN set (slot JI % 10), refilled for the next buffer as soon as the job has
  // consumed it. 10 appends + 18 requests + input / output sample per set in flight at most.
  f4 sa[NH], sb[NH];
  float inp = 0.0f;

  constexpr int kInAux = PERSIST ? 17 : 0; // session inputs bypass the caches (the caller may rewrite the buffer between commands)
  int* const cmd_lds = reinterpret_cast<int*>(lds + p3::kFlagB); // [0] offset, [1] ready: wave 0's decision for the next round
  auto ring_load = [&](unsigned s_) {
    return __hip_atomic_load(a.p_ring + (s_ & (unsigned)a.p_ring_mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  };
  unsigned done = 0; // PERSIST: commands whose buffers are complete (set B finished them)
  auto leave = [&](bool fence) {
    if (fence)
    {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_barrier();
    }
    if (w8 == 0 && lane == 0)
    {
      a.p_cons[blockIdx.x] = done;
      __hip_atomic_store(a.p_done + blockIdx.x, done | 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  };

  // every ring request of the first buffer (they depend on the state only, not on the command): each set its own jobs
  if (S == 0)
  {
    il::for_each_index(
      [&](auto u_tag) {
        constexpr int U = decltype(u_tag)::value;
        fetch(sa[U], sb[U], std::integral_constant<int, U>{}, std::integral_constant<int, 0>{}, true, t, v_g16);
      },
      std::make_integer_sequence<int, NH>{});
  }
  else
  {
    il::for_each_index(
      [&](auto u_tag) {
        constexpr int U = decltype(u_tag)::value;
        fetch(sa[U], sb[U], std::integral_constant<int, NH + U>{}, std::integral_constant<int, 0>{}, true, t, v_g16);
      },
      std::make_integer_sequence<int, NH>{});
  }

  unsigned na = 0; // PERSIST: commands set A has finished (its current command carries tag na + 1)
  unsigned boff_a = 0, boff_b = 0; // byte offsets of set A's / set B's buffer in the stream's row
  if constexpr (PERSIST)
  {
    const bool by_value = a.p_seq0 >= 0;
    done = na = by_value ? (unsigned)a.p_seq0 : a.p_cons[blockIdx.x];
    bool ready = true;
    unsigned lo = (unsigned)a.p_cmd0;
    if (!by_value)
    {
      if (w8 == 0)
      {
        // started right behind a stream-ordered doorbell on another hardware queue: look for it for a bounded time
        unsigned long long v = ring_load(na);
        if (a.p_grace > 0 && (unsigned)(v >> 32) != na + 1)
        {
          const long long t_end = (long long)wall_clock64() + a.p_grace;
          do
          {
            __builtin_amdgcn_s_sleep(8);
            v = ring_load(na);
          } while ((unsigned)(v >> 32) != na + 1 && (long long)wall_clock64() < t_end);
        }
        if (lane == 0)
        {
          cmd_lds[0] = (int)(unsigned)v;
          cmd_lds[1] = (unsigned)(v >> 32) == na + 1 ? 1 : 0;
        }
      }
      lds_barrier();
      ready = uni(cmd_lds[1]) != 0;
      lo = (unsigned)uni(cmd_lds[0]);
      lds_barrier();
    }
    if (!ready)
    {
      leave(false); // nothing to do (the doorbell this launch was started for has been consumed by its predecessor)
      return;
    }
    boff_a = lo * 4u;
  }
  if (S == 0)
    inp = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_in, t * 4, uni((int)boff_a), kInAux));
  // the weights (requested before the ring rows, so they are here first)
#pragma unroll
  for (int i = 0; i < kT4; i++)
    lds_st4(lds, (unsigned)p3::kTilesB + (unsigned)(i * 512 + tid) * 16u, tl4[i]);
  if (tid < NJ * 16)
    lds_st4(lds, (unsigned)p3::kConstsB + (unsigned)tid * 16u, c0v);
  if (tid < p2::kXt * 64)
    lds_st4(lds, (unsigned)p3::kXtB + (unsigned)tid * 16u, x0v);
  lds_barrier();

  auto early = [](int ji) { return ji > 0 && ji < NJ && ji != NH && p2::kind(ji) == IL_HIST; };
  auto load_ops = [&](Ops& o, auto j_tag) {
    constexpr int JN = decltype(j_tag)::value; // the job whose operands are read
    constexpr unsigned consts_b = p3::kConstsB + JN * 256, tiles_b = p3::kTilesB + JN * 4096;
#pragma unroll
    for (int q = (early(JN) ? 2 : 0); q < 4; q++) // (a history job's tap tiles were read by its predecessor)
      o.t[q] = lds_ld4(lds, v_lane16 + tiles_b + 1024u * q);
    o.bv = lds_ld4(lds, v_g16 + consts_b);
    o.mv = lds_ld4(lds, v_g16 + consts_b + 64u);
    o.b1v = lds_ld4(lds, v_g16 + consts_b + 128u);
  };
  auto load_extra = [&](f4& xt, f4& ev, auto j_tag) {
    constexpr int JN = decltype(j_tag)::value;
    xt = lds_ld4(lds, v_lane16 + (unsigned)(p3::kXtB + p2::xt_index(JN) * 1024));
    ev = lds_ld4(lds, v_g16 + (unsigned)(p3::kConstsB + JN * 256) + 192u);
  };

  f4 x = {0.f, 0.f, 0.f, 0.f}, head = {0.f, 0.f, 0.f, 0.f};
  int nvalid = min(kBlock, a.n_frames); // frames of the buffer this wave's set is working on
  float cond = 0.0f;
  Ops O;
  f4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = {0.f, 0.f, 0.f, 0.f}; // the next history job's tap products (see p2: `early`)
  unsigned long long spec_cmd = 0; // PERSIST, set A: this wave's early look at the next command ...
  float inp_spec = 0.0f; // ... and the input sample it requested on a hit
  bool more = false; // this wave's set has another buffer behind the current one (its slots are refilled for it)
  int blk = 0; // buffers this wave's set has finished
  unsigned boff = 0; // byte offset of the current buffer of this wave's set

  // one job, everything about it known at compile time (kernel_a1_p2.hip: `job`; differences are marked p3)
  auto job = [&](auto j_tag) {
    constexpr int JI = decltype(j_tag)::value;
    constexpr int SL = JI % NH; // p3: the request slot
    constexpr IlDesc J = p2::desc(C0, C1, 0, JI);
    constexpr int flags = J.flags;
    constexpr int NK = (flags & CD_HALF) ? 2 : 4;
    constexpr unsigned g16max = (unsigned)J.gp;
    const int act = a.act; // (only read by the run-time-dispatch instantiation)
    __builtin_amdgcn_sched_barrier(0);
    int tl = t;
    unsigned gl16 = v_g16;
    asm volatile("" : "+v"(tl), "+v"(gl16));
    load_ops(O, j_tag);
    constexpr bool kNextEarly = early(JI + 1), kThisEarly = early(JI);
    f4 nt0 = {0.f, 0.f, 0.f, 0.f}, nt1 = {0.f, 0.f, 0.f, 0.f}; // tap tiles of the history job behind this one
    if constexpr (kNextEarly)
    {
      constexpr unsigned ntiles_b = p3::kTilesB + (JI + 1) * 4096;
      nt0 = lds_ld4(lds, v_lane16 + ntiles_b);
      nt1 = lds_ld4(lds, v_lane16 + ntiles_b + 1024u);
    }
    f4 xt = {0.f, 0.f, 0.f, 0.f}, ev = {0.f, 0.f, 0.f, 0.f};
    if constexpr ((flags & (CD_X0 | CD_PRE_HEAD | CD_POST_RECH | CD_POST_OUT)) != 0)
      load_extra(xt, ev, j_tag);
    const f4 Sa = sa[SL], Sb = sb[SL];
    if constexpr (kThisEarly)
      ; // (consumed by the previous job)
    else if constexpr (J.kind == IL_EXCH)
      asm volatile("" ::"v"(Sa));
    else
      asm volatile("" ::"v"(Sa), "v"(Sb)); // one wait for the whole slot (the oldest requests in flight)
    if constexpr ((flags & CD_X0) != 0)
    {
      cond = inp; // this buffer's input sample (requested a buffer ago)
      if constexpr (!PERSIST) // next block's (offset beyond the launch's frames -> 0)
        inp = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_in, tl * 4, uni((blk + 1) * (kBlock * 4)), 0));
      x = ev * cond;
      head = f4{0.f, 0.f, 0.f, 0.f};
    }
    // this job's input -> its history ring
    {
      const unsigned v = (unsigned)(__builtin_amdgcn_readlane(wposv, J.ring_id) + tl);
      const unsigned widx = min(v, v - (unsigned)J.R);
      const bool ok = tl < nvalid && gl16 <= g16max;
      const unsigned off = ok ? __umul24(widx, (unsigned)J.row_b) + gl16 + (unsigned)J.ring_b : kOob;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(il::u4, x), rsrc, (int)off, 0, WT ? 17 : 0);
    }
    // the same job of this set's NEXT buffer: its requests go into the slot just consumed
    fetch(sa[SL], sb[SL], j_tag, std::integral_constant<int, 1>{}, more, tl, gl16);
    // p3, persistent session, set A: the next command. Every A wave looks at the ring in job 4 and, when the command is
    // already there, requests the next buffer's input sample from it in job 6 (unconditional load, out-of-range offset
    // on a miss: the memory-operation pattern of a job never depends on data).
    if constexpr (PERSIST && JI == 4)
      spec_cmd = ring_load(na + 1);
    if constexpr (PERSIST && JI == 6)
    {
      const bool hit = (unsigned)(spec_cmd >> 32) == na + 2;
      const int soff = uni(hit ? (int)((unsigned)spec_cmd * 4u) : 0);
      inp_spec = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_in, hit ? tl * 4 : (int)kOob, soff, kInAux));
    }
    auto slice = [&](const f4& r) { return NK == 4 ? r : (hi_pair ? f4{r[2], r[3], 0.f, 0.f} : f4{r[0], r[1], 0.f, 0.f}); };
    f4 bt0, bt1;
    if constexpr (J.kind == IL_HIST)
    {
      bt0 = slice(Sa);
      bt1 = slice(Sb);
    }
    else if constexpr (J.kind == IL_DPP)
    {
      bt0 = bt1 = f4{0.f, 0.f, 0.f, 0.f};
      il::dpp_taps<NK, J.dil / 4>(x, slice(Sa), slice(Sb), bt0, bt1);
    }
    else
    {
      // exchange: jobs 0, 1 -> windows 0, 1 (set A); jobs 10, 11 -> windows 2, 3 (set B). Row order as in p2.
      constexpr unsigned kRowB = 80u;
      auto win_off = [&](unsigned F) { return ((F & 64u) + ((F & 3u) << 4) + ((F & 63u) >> 2)) * kRowB; };
      constexpr unsigned wb = (unsigned)((JI & 1) + (JI >= NH ? 2 : 0)) * (unsigned)p3::kWinB;
      if (gl16 <= g16max)
      {
        lds_st4(lds, wb + win_off((unsigned)(kBlock + tl)) + gl16, x);
        lds_st4(lds, wb + win_off((unsigned)tl) + gl16, Sa);
      }
      lds_barrier();
      const unsigned chan = NK == 4 ? min(gl16, g16max) : v_gh8;
      const unsigned r1 = wb + win_off((unsigned)(kBlock + tl - J.dil)) + chan;
      const unsigned r0 = wb + win_off((unsigned)(kBlock + tl - 2 * J.dil)) + chan;
      if constexpr (NK == 4)
      {
        bt1 = lds_ld4(lds, r1);
        bt0 = lds_ld4(lds, r0);
      }
      else
      {
        const f2 p1 = *reinterpret_cast<const f2*>(lds + r1);
        const f2 p0 = *reinterpret_cast<const f2*>(lds + r0);
        bt1 = f4{p1[0], p1[1], 0.f, 0.f};
        bt0 = f4{p0[0], p0[1], 0.f, 0.f};
      }
    }
    if constexpr ((flags & CD_PRE_HEAD) != 0)
      head = ((flags & CD_PREV_HALF) ? mfma_n<2>(xt, head, f4{0.f, 0.f, 0.f, 0.f}) : mfma_n<4>(xt, head, f4{0.f, 0.f, 0.f, 0.f}))
             + ev;
    f4 acc0 = O.mv * cond, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = O.bv;
#pragma unroll
    for (int s = 0; s < NK; s++)
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(O.t[2][s], x[s], acc2, 0, 0, 0);
    if constexpr (kThisEarly)
    {
      acc0 += e0; // computed by the previous job
      acc1 = e1;
    }
    else
    {
#pragma unroll
      for (int s = 0; s < NK; s++)
      {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(O.t[0][s], bt0[s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(O.t[1][s], bt1[s], acc1, 0, 0, 0);
      }
    }
    if constexpr (kNextEarly)
    {
      constexpr int NKN = (p2::desc(C0, C1, 0, JI + 1 < NJ ? JI + 1 : JI).flags & CD_HALF) ? 2 : 4;
      constexpr int SN = (JI + 1) % NH;
      const f4 Na = sa[SN], Nb = sb[SN];
      asm volatile("" ::"v"(Na), "v"(Nb));
      auto nslice = [&](const f4& r) { return NKN == 4 ? r : (hi_pair ? f4{r[2], r[3], 0.f, 0.f} : f4{r[0], r[1], 0.f, 0.f}); };
      const f4 n0 = nslice(Na), n1 = nslice(Nb);
      e0 = f4{0.f, 0.f, 0.f, 0.f};
      e1 = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NKN; s++)
      {
        e0 = __builtin_amdgcn_mfma_f32_16x16x4f32(nt0[s], n0[s], e0, 0, 0, 0);
        e1 = __builtin_amdgcn_mfma_f32_16x16x4f32(nt1[s], n1[s], e1, 0, 0, 0);
      }
    }
    const f4 pre = (acc0 + acc1) + acc2;
    const f4 z = act4<ACT_T>(act, NK == 2 ? f4{pre[0], pre[1], pre[0], pre[1]} : pre, act_p0);
    head += z;
    asm volatile("" : "+v"(head)); // (pin the accumulator: kernel_a1_p2.hip)
    f4 y0 = x + O.b1v, y1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NK; s += 2)
    {
      y0 = __builtin_amdgcn_mfma_f32_16x16x4f32(O.t[3][s], z[s], y0, 0, 0, 0);
      y1 = __builtin_amdgcn_mfma_f32_16x16x4f32(O.t[3][s + 1], z[s + 1], y1, 0, 0, 0);
    }
    x = y0 + y1;
    if constexpr ((flags & CD_POST_OUT) != 0)
    {
      const float yout = head_scale * (mfma_n<NK>(xt, head, f4{0.f, 0.f, 0.f, 0.f}) + ev)[0];
      const bool ok = gl16 == 0 && tl < nvalid;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yout), rsrc_out, ok ? tl * 4 : (int)kOob,
                                            uni((int)boff), PERSIST ? 17 : 0);
    }
    else if constexpr ((flags & CD_POST_RECH) != 0)
      x = mfma_n<NK>(xt, x, f4{0.f, 0.f, 0.f, 0.f});
  };

  // hand-over buffer `par` (0 / 1): lane l of wave w of set A -> lane l of wave w of set B
  auto hand_b = [&](int par) { return (unsigned)p3::kHandB + (unsigned)par * (unsigned)p3::kHandBufB; };

  // ---- rounds ----
  // Each set runs its OWN round loop (one straight-line loop body per set: the compiler's wait-count bookkeeping sees a
  // plain loop of ten jobs, as in nam_a1_p2_kernel); both keep the same workgroup-uniform books — which set has a buffer
  // in this round, commands started / finished, buffer offsets — and meet at the same three barriers per round.
  auto run = [&](auto set_tag) {
    constexpr int SET = decltype(set_tag)::value;
    bool have_a = n_blocks > 0, have_b = false; // does set A / set B have a buffer in this round (workgroup-uniform)
    int blocks_started = 0; // buffers set A has started or finished (non-persistent: the next one is block `blocks_started`)
#pragma unroll 1
    for (;;)
    {
      if constexpr (SET == 0)
      {
        if (have_a)
        {
          // is there a buffer behind this one? persistent: unknown yet — request anyway, the rows exist
          more = PERSIST || blocks_started + 1 < n_blocks;
          boff = boff_a;
          nvalid = PERSIST ? kBlock : min(kBlock, a.n_frames - blocks_started * kBlock);
          il::for_each_index(job, std::make_integer_sequence<int, NH>{});
          // hand-over: x (array 1's rechannelled input), array 0's head accumulator, the input sample
          const unsigned hb = hand_b(blk & 1);
          lds_st4(lds, hb + (unsigned)tid_s * 16u, x);
          lds_st4(lds, hb + 4096u + (unsigned)tid_s * 16u, head);
          *reinterpret_cast<float*>(lds + hb + 8192u + (unsigned)tid_s * 4u) = cond;
          wposv += nvalid;
          if (wposv >= ring_len_v)
            wposv -= ring_len_v;
          blk++;
        }
        else
        {
          lds_barrier(); // the exchange barriers of set B's jobs 10 / 11
          lds_barrier();
        }
        // wave 0 decides about the next round's buffer for set A
        if constexpr (PERSIST)
        {
          if (w8 == 0)
          {
            const unsigned tag = na + (have_a ? 2u : 1u); // the command behind the one just finished (or the one still missing)
            unsigned long long v = have_a ? spec_cmd : ring_load(tag - 1u);
            if ((unsigned)(v >> 32) != tag)
            {
              // missed in job 4 / idle round: look again — while set B drains, for a few microseconds (bounded: the
              // launch never waits for a command)
              v = ring_load(tag - 1u);
              if (!have_a && have_b)
              {
                const long long t_end = (long long)wall_clock64() + 300; // 3 us of the 100 MHz clock
                while ((unsigned)(v >> 32) != tag && (long long)wall_clock64() < t_end)
                {
                  __builtin_amdgcn_s_sleep(16);
                  v = ring_load(tag - 1u);
                }
              }
            }
            if (lane == 0)
            {
              cmd_lds[0] = (int)(unsigned)v;
              cmd_lds[1] = (unsigned)(v >> 32) == tag ? 1 : 0;
            }
          }
        }
      }
      else
      {
        if (have_b)
        {
          more = PERSIST || have_a; // set A is on the buffer behind this one (a session refills anyway: the rows exist,
                                    // and a command that arrives in the drain round finds the slots ready)
          boff = boff_b;
          nvalid = PERSIST ? kBlock : min(kBlock, a.n_frames - blk * kBlock);
          const unsigned hb = hand_b(blk & 1);
          x = lds_ld4(lds, hb + (unsigned)tid_s * 16u);
          head = lds_ld4(lds, hb + 4096u + (unsigned)tid_s * 16u);
          cond = *reinterpret_cast<const float*>(lds + hb + 8192u + (unsigned)tid_s * 4u);
          il::for_each_index([&](auto u_tag) { job(std::integral_constant<int, NH + decltype(u_tag)::value>{}); },
                             std::make_integer_sequence<int, NH>{});
          wposv += nvalid;
          if (wposv >= ring_len_v)
            wposv -= ring_len_v;
          blk++;
        }
        else
        {
          lds_barrier(); // the exchange barriers of set A's jobs 0 / 1
          lds_barrier();
        }
      }
      lds_barrier(); // round boundary: the hand-over and wave 0's decision are the workgroup's
      bool next_ready;
      unsigned next_off;
      if constexpr (PERSIST)
      {
        next_ready = uni(cmd_lds[1]) != 0;
        next_off = (unsigned)uni(cmd_lds[0]) * 4u;
        if (have_b)
        {
          done++;
          if (w8 == 4 && lane == 0 && (done & 15u) == 0u) // progress for the host's ring bookkeeping (not a completion signal)
            __hip_atomic_store(a.p_prog + blockIdx.x, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (have_a)
          na++;
      }
      else
      {
        if (have_a)
          blocks_started++;
        next_ready = blocks_started < n_blocks;
        next_off = (unsigned)blocks_started * (kBlock * 4u);
      }
      const bool a_ran = have_a;
      boff_b = boff_a;
      have_b = have_a;
      have_a = next_ready;
      if (!have_a && !have_b)
        break;
      if (have_a)
      {
        boff_a = next_off;
        if constexpr (PERSIST && SET == 0)
        {
          // the input sample of the buffer about to start: requested in job 6 when this wave saw the command itself
          const bool mine = a_ran && (unsigned)(spec_cmd >> 32) == na + 1u && (unsigned)spec_cmd * 4u == next_off;
          inp = inp_spec;
          if (!mine)
          {
            // The rare path (this wave's look in job 4 came too early, or set A idled): load now and wait right here,
            // INSIDE the asm. A load the compiler can see at this point would be the youngest operation in flight when
            // job 0 consumes it, and its wait-count pass would drain every ring request of the round (vmcnt(0)) in
            // front of job 0 on every round, taken or not.
            const int voff = t * 4, soff = uni((int)next_off);
            const i4 rs = in_desc; // (the descriptor of rsrc_in as four dwords: an asm operand)
            asm volatile("buffer_load_dword %0, %1, %2, %3 offen sc0 sc1\n\ts_waitcnt vmcnt(0)"
                         : "=v"(inp)
                         : "v"(voff), "s"(rs), "s"(soff)
                         : "memory");
          }
        }
      }
      // (cmd_lds is rewritten by wave 0 at the end of the coming round, behind its two exchange barriers: every wave has
      // read it by then)
    }
  };
  if (S == 0)
    run(std::integral_constant<int, 0>{});
  else
    run(std::integral_constant<int, 1>{});
  if (w == 0 && lane >= S * NH && lane < S * NH + NH)
    wpos_tbl[lane] = wposv;
  if constexpr (PERSIST)
    leave(true);
}

namespace
{
template <int C0, int C1, int ACT_T, bool WT, bool PERSIST = false>
hipError_t launch_p3_inst(const A1Args& a, int n_blocks, hipStream_t stream)
{
  static DynamicLdsLimit lds_limit; // per instantiation, tracked per device (kernels.h)
  const hipError_t e = lds_limit.ensure(reinterpret_cast<const void*>(&nam_a1_p3_kernel<C0, C1, ACT_T, WT, PERSIST>), p3::kLdsBytes);
  if (e != hipSuccess)
    return e;
  hipLaunchKernelGGL((nam_a1_p3_kernel<C0, C1, ACT_T, WT, PERSIST>), dim3(n_blocks), dim3(512), p3::kLdsBytes, stream, a.blob, a);
  return hipGetLastError();
}
template <int C0, int C1>
hipError_t launch_p3_shape(const A1Args& a, int n_blocks, int act, hipStream_t stream)
{
  if (a.p_ring) // persistent session: write-back ring appends (kernel_a1_p2.hip: launch_p2_shape)
  {
    if (act == ACT_FASTTANH)
      return launch_p3_inst<C0, C1, ACT_FASTTANH, false, true>(a, n_blocks, stream);
    if (act == ACT_TANH)
      return launch_p3_inst<C0, C1, ACT_TANH, false, true>(a, n_blocks, stream);
    return launch_p3_inst<C0, C1, -1, false, true>(a, n_blocks, stream);
  }
  const bool wt = a.n_frames <= 2 * kBlock; // short launches write ring appends through (device_common.h: ring_store)
  if (act == ACT_FASTTANH)
    return wt ? launch_p3_inst<C0, C1, ACT_FASTTANH, true>(a, n_blocks, stream) : launch_p3_inst<C0, C1, ACT_FASTTANH, false>(a, n_blocks, stream);
  if (act == ACT_TANH)
    return wt ? launch_p3_inst<C0, C1, ACT_TANH, true>(a, n_blocks, stream) : launch_p3_inst<C0, C1, ACT_TANH, false>(a, n_blocks, stream);
  return wt ? launch_p3_inst<C0, C1, -1, true>(a, n_blocks, stream) : launch_p3_inst<C0, C1, -1, false>(a, n_blocks, stream);
}
} // namespace

hipError_t launch_a1_p3(const A1Args& a, int n_blocks, int c0, int c1, int act, hipStream_t stream)
{
  if (c0 == 16 && c1 == 8)
    return launch_p3_shape<16, 8>(a, n_blocks, act, stream);
  if (c0 == 12 && c1 == 8)
    return launch_p3_shape<12, 8>(a, n_blocks, act, stream);
  if (c0 == 8 && c1 == 4)
    return launch_p3_shape<8, 4>(a, n_blocks, act, stream);
  return hipErrorInvalidValue;
}

} // namespace namhip
