// kernel_a1_p2.hip — nam_a1_p2_kernel: the interleaved-frame MFMA kernel with the official A1 topology compiled in.
#include "device_common.h"
#include "il_common.h"

namespace namhip
{

// ================================================================================================
// nam_a1_p2_kernel<C0, C1> — the interleaved-frame mapping (plan.h, "Interleaved-frame MFMA kernel": frames t = 4 j + w per
// compute wave, DPP / ring / exchange jobs, request slots) for ONE topology: two arrays of ten layers, kernel
// size 3, dilations 1 ... 512 — every official WaveNet size (standard 16 / 8 channels, lite 12 / 6 -> 8, feather
// 8 / 4). The job table is plan.h's constexpr p2::desc / p2::fetch evaluated at compile time (plan_a1.cpp only selects
// this kernel when those functions reproduce the model's run-time tables bit for bit), so a block is twenty
// straight-line jobs: no descriptor loads, no kind / layout / flag tests, immediate LDS offsets and ring constants.
// Why it exists: with one compute wave per SIMD the INSTRUCTION COUNT is the time. The descriptor-driven kernel
// issues ~340 instructions per job for 16 MFMAs (2,100 cycles measured); this one ~1/3 of that.
// Four waves per stream, no loader wave: the weights are staged in LDS by the compute waves themselves in front of
// the first job (they arrive while the ring requests are in flight anyway), which leaves ONE wave per SIMD — the whole
// 512-entry register file per wave — and that is what allows a request slot for every job of a block.
// Same state layout, rings and write positions as every other A1 kernel; same numerics (the MFMAs and the
// activation code are the same, only the control flow is resolved by the compiler).
// ================================================================================================
// PERSIST (block mode without a kernel boundary per buffer): the launch consumes COMMANDS — one per 64-frame buffer,
// `(seq << 32) | frame offset`, written into a device-memory ring by hipStreamWriteValue64 on the caller's stream
// (api_session.cpp: persistent session) — for as long as the next one is already there. Wave 0 looks at command k + 1
// while block k is still computing and the workgroup agrees on it at the end of the block (one extra barrier). With
// the ring empty the workgroup makes its results visible (system-scope release), publishes how many commands it has
// consumed and LEAVES: it never waits, so nothing can hang and a device-wide synchronise simply returns when the
// buffers rung so far are done; the host starts the next launch with the next doorbell. While doorbells arrive faster
// than blocks are computed (3-6 us vs 10 us) a whole run of buffers is one launch: dispatch, kernarg and
// write-position round trips, weights into LDS and the write-through drain are paid once per run, not per buffer.
template <int C0, int C1, int ACT_T, bool WT, bool PERSIST>
__global__ __launch_bounds__(256) void nam_a1_p2_kernel(const float* __restrict__ blob, const A1Args a)
{
  using namespace mf;
  using il::kOob;
  using il::Ops;
  constexpr int NJ = p2::kJobs;
  extern __shared__ __attribute__((aligned(16))) float lds_p2[];
  char* const lds = reinterpret_cast<char*>(lds_p2);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = uni(tid >> 6);
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
  // (persistent sessions address a whole resident window of the stream's row: commands carry frame offsets into it)
  const int io_bytes = PERSIST ? 0x7ffffff0 : a.n_frames * 4;
  const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc((void*)(in ? in : st), 0, in ? io_bytes : 0, 0x00020000);
  const auto rsrc_out = __builtin_amdgcn_make_buffer_rsrc((void*)(out ? out : st), 0, out ? io_bytes : 0, 0x00020000);
  int* wpos_tbl = reinterpret_cast<int*>(st);
  int wposv = lane < NJ ? wpos_tbl[lane] : 0; // lane r = write position of ring r (ring r = job r)
  const int ring_len_v = 2 * (1 << (lane % p2::kLayers)) + kBlock;
  int blk = 0;

  // ---- weights -> LDS, once per launch, by all four waves (no loader wave: with one wave per SIMD each wave may use
  // the whole 512-entry register file, which is what lets a full block of ring requests stay in flight) ----
  // requested first, written to LDS after the ring requests below have been issued behind them
  constexpr int kT4 = NJ * 256 / 256; // 16-byte tile records per thread: 20 jobs x 256 records / 256 threads
  f4 tl4[kT4];
  {
    const f4* __restrict__ tsrc = reinterpret_cast<const f4*>(blob + a.tiles_off);
#pragma unroll
    for (int i = 0; i < kT4; i++)
      tl4[i] = tsrc[i * 256 + tid];
  }
  const f4* __restrict__ csrc = reinterpret_cast<const f4*>(blob + a.consts_off);
  const f4* __restrict__ xsrc = reinterpret_cast<const f4*>(blob + a.xt_off);
  const f4 c0v = csrc[tid], c1v = csrc[min(tid + 256, NJ * 16 - 1)]; // 320 constant records
  const f4 x0v = xsrc[min(tid, p2::kXt * 64 - 1)]; // 192 extra-tile records

  // the ring requests of job TJ, which belongs to block blk + AHEAD; (tl / gl16: the lane's frame and channel-quad
  // offset, laundered per job by the caller — see `job`)
  auto fetch = [&](f4& sa, f4& sb, auto tj_tag, auto ahead_tag, bool valid, int tl, unsigned gl16) {
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
          sa = r;
        else
          sb = r;
      }
    }
  };
  // Request slots: ONE PER JOB of a block (a job's slot is refilled for the next block as soon as the job has consumed
  // it), so a launch of one block has every ring request in flight before its first job starts — one memory round
  // trip per launch — and a resident launch looks a whole block ahead. 20 appends + 36 requests + input + output
  // sample = 58 vector-memory operations in flight at most (the counter holds 63).
  f4 sa[NJ], sb[NJ];
  float inp;
  auto prologue = [&](auto u_tag) {
    constexpr int U = decltype(u_tag)::value;
    fetch(sa[U], sb[U], std::integral_constant<int, U>{}, std::integral_constant<int, 0>{}, true, t, v_g16);
  };
  // ---- persistent session: command ring -------------------------------------------------------------------
  // Input samples of a persistent session bypass L1 / L2 (sc0 sc1): the caller may rewrite the same input buffer between
  // two commands (the blocking host path copies every buffer into one staging area), and without a kernel boundary
  // nothing invalidates the line this CU read a buffer ago.
  constexpr int kInAux = PERSIST ? 17 : 0;
  unsigned seq = 0; // commands consumed so far
  unsigned boff = 0; // byte offset of the current block's frames in the stream's row (non-persistent: blk * 256)
  int* const cmd_lds = reinterpret_cast<int*>(lds_p2) + p2::kFlagB / 4; // [0..1] = command agreed on by the workgroup
  auto ring_load = [&](unsigned s_) {
    return __hip_atomic_load(a.p_ring + (s_ & (unsigned)a.p_ring_mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  };
  // wave 0 waits (bounded) for command `want`; returns its low word, or kExit on EXIT / expiry
  // the workgroup leaves: results visible, then its consumed-command count (device copy for the next launch of this
  // workgroup, host copy for the host), exited bit set
  // (The session's output samples are stored write-through at system scope, so "every store of this wave has been
  // acknowledged" is all the count has to wait for: a release fence here would first write back the L2's dirty
  // history rows — megabytes that only the next launch needs, and the end of this one takes care of those.)
  auto leave = [&](bool fence) {
    if (fence)
    {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_barrier();
    }
    if (w == 0 && lane == 0)
    {
      a.p_cons[blockIdx.x] = seq;
      __hip_atomic_store(a.p_done + blockIdx.x, seq | 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  };
  // every ring request of the first block (they depend on the state only, not on the command)
  il::for_each_index(prologue, std::make_integer_sequence<int, NJ>{});
  if constexpr (PERSIST)
  {
    // commands this workgroup has consumed in earlier launches of the session — by value when the host knows that
    // every workgroup stands at the same count (p_seq0 >= 0), and then the first command comes by value as well
    const bool by_value = a.p_seq0 >= 0;
    seq = by_value ? (unsigned)a.p_seq0 : a.p_cons[blockIdx.x];
    bool ready = true;
    unsigned lo = (unsigned)a.p_cmd0;
    if (!by_value)
    {
      if (w == 0)
      {
        // The host starts this launch right behind the doorbell it rang, on another hardware queue: the doorbell may
        // land a few microseconds after the launch. Wave 0 looks for it for a bounded time (a.p_grace ticks of the
        // 100 MHz clock; never unbounded: a device-wide synchronize must not depend on a doorbell being delivered).
        unsigned long long v = ring_load(seq);
        if (a.p_grace > 0 && (unsigned)(v >> 32) != seq + 1)
        {
          const long long t_end = (long long)wall_clock64() + a.p_grace;
          do
          {
            __builtin_amdgcn_s_sleep(8);
            v = ring_load(seq);
          } while ((unsigned)(v >> 32) != seq + 1 && (long long)wall_clock64() < t_end);
        }
        if (lane == 0)
        {
          cmd_lds[0] = (int)(unsigned)v;
          cmd_lds[1] = (unsigned)(v >> 32) == seq + 1 ? 1 : 0;
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
    boff = lo * 4u;
  }
  inp = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_in, t * 4, uni((int)boff), kInAux));
  // the weights (requested before the ring rows, so they are here first)
#pragma unroll
  for (int i = 0; i < kT4; i++)
    lds_st4(lds, (unsigned)p2::kTilesB + (unsigned)(i * 256 + tid) * 16u, tl4[i]);
  lds_st4(lds, (unsigned)p2::kConstsB + (unsigned)tid * 16u, c0v);
  if (tid + 256 < NJ * 16)
    lds_st4(lds, (unsigned)p2::kConstsB + (unsigned)(tid + 256) * 16u, c1v);
  if (tid < p2::kXt * 64)
    lds_st4(lds, (unsigned)p2::kXtB + (unsigned)tid * 16u, x0v);
  lds_barrier();

  // History jobs (dilation >= 64: both shifted taps are pure history, requested a block ago) get their two tap products
  // from the job BEFORE them: those eight MFMAs run in the matrix pipe while that job's activation occupies the vector
  // pipe, and the history job's own critical path shrinks to the current-frame chain (plan.h: p2::kind).
  auto early = [](int ji) { return ji > 0 && ji < NJ && p2::kind(ji) == IL_HIST; };
  auto load_ops = [&](Ops& o, auto j_tag) {
    constexpr int JN = decltype(j_tag)::value; // the job whose operands are read
    constexpr unsigned consts_b = p2::kConstsB + JN * 256, tiles_b = p2::kTilesB + JN * 4096;
#pragma unroll
    for (int q = (early(JN) ? 2 : 0); q < 4; q++) // (a history job's tap tiles were read by its predecessor)
      o.t[q] = lds_ld4(lds, v_lane16 + tiles_b + 1024u * q);
    o.bv = lds_ld4(lds, v_g16 + consts_b);
    o.mv = lds_ld4(lds, v_g16 + consts_b + 64u);
    o.b1v = lds_ld4(lds, v_g16 + consts_b + 128u);
  };
  // extra tile + extra constants of job JN (array entry / exit jobs only)
  auto load_extra = [&](f4& xt, f4& ev, auto j_tag) {
    constexpr int JN = decltype(j_tag)::value;
    xt = lds_ld4(lds, v_lane16 + (unsigned)(p2::kXtB + p2::xt_index(JN) * 1024));
    ev = lds_ld4(lds, v_g16 + (unsigned)(p2::kConstsB + JN * 256) + 192u);
  };

  f4 x = {0.f, 0.f, 0.f, 0.f}, head = {0.f, 0.f, 0.f, 0.f};
  int nvalid = min(kBlock, a.n_frames);
  float cond = 0.0f;
  // One operand register set: a job reads its own tiles / constants from LDS when it starts and hides the latency
  // behind its ring append, its requests and its tap shuffles.
  Ops O;

  f4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = {0.f, 0.f, 0.f, 0.f}; // the next history job's tap products (see `early`)
  unsigned long long spec_cmd = 0; // persistent: this wave's early look at the next command
  bool spec_ok = false;
  unsigned spec_off = 0;
  // one job, everything about it known at compile time
  auto job = [&](auto j_tag) {
    constexpr int JI = decltype(j_tag)::value;
    constexpr IlDesc J = p2::desc(C0, C1, 0, JI);
    constexpr int flags = J.flags;
    constexpr int NK = (flags & CD_HALF) ? 2 : 4;
    constexpr unsigned g16max = (unsigned)J.gp;
    const int act = a.act; // (only read by the run-time-dispatch instantiation)
    __builtin_amdgcn_sched_barrier(0);
    // The lane's frame and quad offset, opaque to the optimiser from here on: every address below would otherwise be a
    // block-loop invariant, computed once for all twenty jobs in front of the loop and kept in ~40 VGPRs; recomputing
    // costs two or three VALU instructions per use.
    int tl = t;
    unsigned gl16 = v_g16;
    asm volatile("" : "+v"(tl), "+v"(gl16));
    load_ops(O, j_tag);
    constexpr bool kNextEarly = early(JI + 1), kThisEarly = early(JI);
    f4 nt0 = {0.f, 0.f, 0.f, 0.f}, nt1 = {0.f, 0.f, 0.f, 0.f}; // tap tiles of the history job behind this one
    if constexpr (kNextEarly)
    {
      constexpr unsigned ntiles_b = p2::kTilesB + (JI + 1) * 4096;
      nt0 = lds_ld4(lds, v_lane16 + ntiles_b);
      nt1 = lds_ld4(lds, v_lane16 + ntiles_b + 1024u);
    }
    f4 xt = {0.f, 0.f, 0.f, 0.f}, ev = {0.f, 0.f, 0.f, 0.f};
    if constexpr ((flags & (CD_X0 | CD_PRE_HEAD | CD_POST_RECH | CD_POST_OUT)) != 0)
      load_extra(xt, ev, j_tag);
    const f4 Sa = sa[JI], Sb = sb[JI];
    if constexpr (kThisEarly)
      ; // (consumed by the previous job)
    else if constexpr (J.kind == IL_EXCH)
      asm volatile("" ::"v"(Sa));
    else
      asm volatile("" ::"v"(Sa), "v"(Sb)); // one wait for the whole slot (the oldest requests in flight)
    if constexpr ((flags & CD_X0) != 0)
    {
      cond = inp; // this block's input sample (requested a block ago)
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
    // the same job of the NEXT block: its requests go into the slot just consumed
    fetch(sa[JI], sb[JI], j_tag, std::integral_constant<int, 1>{}, PERSIST || blk + 1 < n_blocks, tl, gl16);
    // Persistent session, the NEXT block's command (steady state: no barrier and no exposed load of its own):
    // every wave looks at the ring in job 4; in job 9 wave 0 publishes what it saw in LDS; the exchange barriers of jobs
    // 10 / 11 make that the whole workgroup's view, read back in job 12 — where the next block's input sample is
    // requested from the offset everybody agreed on. (cmd_lds[0..1] is rewritten in job 9 of the next block, behind
    // the barriers of its jobs 0 / 1.)
    if constexpr (PERSIST && JI == 4)
      spec_cmd = ring_load(seq + 1);
    if constexpr (PERSIST && JI == 9)
    {
      const bool hit = (unsigned)(spec_cmd >> 32) == seq + 2;
      if (w == 0 && lane == 0)
      {
        cmd_lds[0] = hit ? (int)(unsigned)spec_cmd : 0;
        cmd_lds[1] = hit ? 1 : 0;
      }
    }
    if constexpr (PERSIST && JI == 12)
    {
      spec_ok = uni(cmd_lds[1]) != 0;
      spec_off = (unsigned)uni(cmd_lds[0]) * 4u;
      inp = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_in, tl * 4, uni((int)spec_off), kInAux));
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
      // exchange: jobs 0, 1, 10, 11 -> windows 0, 1, 0, 1 of every block.
      // Window rows are NOT in frame order: frame F (0 .. 63 previous block, 64 .. 127 this block) sits in row
      // (F & 64) + 16 (F & 3) + ((F & 63) >> 2), rows of 80 bytes. A wavefront's lanes hold frames 4 apart, so in frame
      // order (and any 16-byte-aligned pitch) the sixteen lanes of a b128 phase fell into two bank groups — 8-way
      // conflicts, 2.5 k LDS cycles per block and CU (SQ_LDS_BANK_CONFLICT), all of it in front of a barrier or of the
      // first MFMA. This way they sit in consecutive rows, and 5 sixteen-byte slots per row walk all 16 bank groups.
      constexpr unsigned kRowB = 80u;
      static_assert(2 * kBlock * kRowB <= (unsigned)kIlWinB, "p2 exchange window");
      auto win_off = [&](unsigned F) { return ((F & 64u) + ((F & 3u) << 4) + ((F & 63u) >> 2)) * kRowB; };
      constexpr unsigned wb = (unsigned)(JI & 1) * (unsigned)kIlWinB;
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
      // the history job behind this one: its two shifted taps (slot JI + 1, requested a block ago) times its tap tiles
      constexpr int NKN = (p2::desc(C0, C1, 0, JI + 1 < NJ ? JI + 1 : JI).flags & CD_HALF) ? 2 : 4;
      const f4 Na = sa[JI + 1 < NJ ? JI + 1 : JI], Nb = sb[JI + 1 < NJ ? JI + 1 : JI];
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
    // pin the accumulator: otherwise the optimiser sinks these adds to the next USE of `head` (ten jobs later) and
    // keeps every job's z alive until then — 40 VGPRs
    asm volatile("" : "+v"(head));
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

#pragma unroll 1
  for (int b = 0; b < n_blocks; b++)
  {
    il::for_each_index(job, std::make_integer_sequence<int, NJ>{});
    wposv += nvalid;
    if (wposv >= ring_len_v)
      wposv -= ring_len_v;
    blk++;
    if constexpr (!PERSIST)
    {
      nvalid = min(kBlock, a.n_frames - blk * kBlock);
      boff = (unsigned)blk * (kBlock * 4u);
    }
    else
    {
      seq++;
      if (w == 0 && lane == 0 && (seq & 15u) == 0u) // progress for the host's ring bookkeeping (no fence: not a completion signal)
        __hip_atomic_store(a.p_prog + blockIdx.x, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (spec_ok) // (the same for every wave: read from LDS behind a barrier)
        boff = spec_off;
      else
      {
        // the command was not there at job 4: look once more — wave 0's view, agreed on through LDS (words 2, 3:
        // another wave may still be on its way to reading words 0, 1 in job 12)
        if (w == 0)
        {
          const unsigned long long v = ring_load(seq);
          const bool ready = (unsigned)(v >> 32) == seq + 1;
          if (lane == 0)
          {
            cmd_lds[2] = ready ? (int)(unsigned)v : 0;
            cmd_lds[3] = ready ? 1 : 0;
          }
        }
        lds_barrier();
        const unsigned lo = (unsigned)uni(cmd_lds[2]);
        const bool ready = uni(cmd_lds[3]) != 0;
        lds_barrier();
        if (!ready)
          break; // ring empty: leave (below)
        boff = lo * 4u;
        inp = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_in, t * 4, uni((int)boff), kInAux));
      }
    }
  }
  if (w == 0 && lane < NJ)
    wpos_tbl[lane] = wposv;
  if constexpr (PERSIST)
    leave(true);
}

namespace
{
template <int C0, int C1, int ACT_T, bool WT, bool PERSIST = false>
hipError_t launch_p2_inst(const A1Args& a, int n_blocks, hipStream_t stream)
{
  static DynamicLdsLimit lds_limit; // per instantiation, tracked per device (kernels.h)
  const hipError_t e = lds_limit.ensure(reinterpret_cast<const void*>(&nam_a1_p2_kernel<C0, C1, ACT_T, WT, PERSIST>), p2::kLdsBytes);
  if (e != hipSuccess)
    return e;
  nam_launch((nam_a1_p2_kernel<C0, C1, ACT_T, WT, PERSIST>), dim3(n_blocks), dim3(256), p2::kLdsBytes, stream, a.blob, a);
  return hipGetLastError();
}
template <int C0, int C1>
hipError_t launch_p2_shape(const A1Args& a, int n_blocks, int act, hipStream_t stream)
{
  // persistent session: plain write-back ring appends (written through, the rows a block appends would be read back
  // from memory instead of the L2 one block later: 27 us per block instead of 10.5, measured)
  if (a.p_ring)
  {
    if (act == ACT_FASTTANH)
      return launch_p2_inst<C0, C1, ACT_FASTTANH, false, true>(a, n_blocks, stream);
    if (act == ACT_TANH)
      return launch_p2_inst<C0, C1, ACT_TANH, false, true>(a, n_blocks, stream);
    return launch_p2_inst<C0, C1, -1, false, true>(a, n_blocks, stream);
  }
  const bool wt = a.n_frames <= 2 * kBlock; // short launches write ring appends through (see ring_store)
  if (act == ACT_FASTTANH)
    return wt ? launch_p2_inst<C0, C1, ACT_FASTTANH, true>(a, n_blocks, stream) : launch_p2_inst<C0, C1, ACT_FASTTANH, false>(a, n_blocks, stream);
  if (act == ACT_TANH)
    return wt ? launch_p2_inst<C0, C1, ACT_TANH, true>(a, n_blocks, stream) : launch_p2_inst<C0, C1, ACT_TANH, false>(a, n_blocks, stream);
  return wt ? launch_p2_inst<C0, C1, -1, true>(a, n_blocks, stream) : launch_p2_inst<C0, C1, -1, false>(a, n_blocks, stream);
}
} // namespace

hipError_t launch_a1_p2(const A1Args& a, int n_blocks, int c0, int c1, int act, hipStream_t stream)
{
  if (c0 == 16 && c1 == 8)
    return launch_p2_shape<16, 8>(a, n_blocks, act, stream);
  if (c0 == 12 && c1 == 8)
    return launch_p2_shape<12, 8>(a, n_blocks, act, stream);
  if (c0 == 8 && c1 == 4)
    return launch_p2_shape<8, 4>(a, n_blocks, act, stream);
  return hipErrorInvalidValue;
}

} // namespace namhip
