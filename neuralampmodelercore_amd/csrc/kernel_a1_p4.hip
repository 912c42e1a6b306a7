// kernel_a1_p4.hip — nam_a1_p4_kernel: the official A1 topology as a PIPELINE OF WAVE SETS (stages), decoupled through LDS.
#include "device_common.h"
#include "il_common.h"

namespace namhip
{

// ================================================================================================
// What nam_a1_p2_kernel (kernel_a1_p2.hip; read its header first) leaves on the table: a block is a chain of 20 dependent
// jobs on ONE wavefront per SIMD — the matrix pipe busy a third of the time, 38 % of the wave cycles issue stalls, 31 %
// waits (profiles/r02/rocprofv3_summary_c2_p2.txt). A WaveNet block has no parallelism left inside a stream (frames are
// already spread over four waves), but CONSECUTIVE buffers of a stream do: layer l of buffer k + 1 only needs layer l - 1
// of buffer k + 1 and its own history. So the twenty jobs of a block are cut into NST stages of consecutive jobs; stage s
// is a set of four waves (frames t = 4 j + w, the lane layout of kernel_a1_p2.hip) working on buffer k while stage s + 1
// works on buffer k - 1. Any cut works: a stage hands the same 36 bytes per lane to the next one — the layer output x
// (model.cpp:355-392), the head accumulator (:513-531) and the input sample (the condition, :839) — lane to lane, because
// wave w of every stage owns the same frames in the same layout. NOTHING is synchronised workgroup-wide while buffers flow:
//   * wave w of stage s + 1 waits for wave w of stage s only: a one-slot queue per wave in LDS (data + a 16-byte token:
//     output offset, valid frames, EXIT), "produced" / "consumed" words with one writer each, polled from inline asm;
//   * the exchange jobs (dilations 1, 2: the only layers whose taps cross waves) synchronise the four waves of their own
//     stage through four generation words in LDS; a stage holds both exchange jobs of an array or none (each one's
//     barrier is what keeps the other one's window from being overwritten early);
//   * the command protocol of the persistent mode lives in stage 0 (every wave looks at the next ring slot in job 1 and,
//     on a hit, requests the next buffer's input sample itself in job 3 — a hit is always THE next command; wave 0
//     decides at the end of the buffer, polling a few microseconds while the later stages still work); the last stage
//     counts finished buffers. With the ring empty stage 0 sends an EXIT token down the pipeline: every stage finishes
//     what is in front of it and leaves — the launch never waits for a command.
// NST = 3 (jobs 0-5 | 6-12 | 13-19): twelve waves per stream, three per SIMD, ~134 VGPRs each — a stage's request
// slots are its own 6-7 jobs'.
// History of the idea (same-box A/B on the headline shape, us per buffer, driver-shaped 20-buffer region / steady state):
// four-wave kernel 11.2 / 9.8; two wave sets (array 0 | array 1) in lock step through three workgroup barriers per
// round 9.5 / 7.9 — the younger set loses the issue arbitration and the older one waits a fifth of its time; stages
// decoupled through LDS, same job bodies: no better (8.0 steady at 2, 3 or 4 stages: the SIMD's vector-issue slots and
// its matrix pipe were each ~43 % busy and would not overlap further — instruction count had become the time); the
// instruction-lean job body below 8.6 / 7.6.
// Same state (rings, write positions), weights and LDS tiles as nam_a1_p2_kernel — the two alternate freely between
// launches of one stream; sums are associated differently (one chain per product): equal to ~1e-6.
// ================================================================================================
// MUBUF with index AND offset registers (address = base + soffset + index * stride + offset): the ring row index goes
// in as it is — no multiply / shift / add per address — and an index beyond num_records makes the access a no-op. clang
// has builtins for the raw form only; the LLVM intrinsics are declared by name (the compiler's wait-count bookkeeping
// covers them like any other vector-memory instruction).
using p4_i4 = __attribute__((ext_vector_type(4))) int;
__device__ mf::f4 p4_sb_load(p4_i4 rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.v4f32");
__device__ void p4_sb_store(mf::f4 v, p4_i4 rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.store.v4f32");

namespace p4
{
constexpr int kNoRow = 1 << 26; // a ring row index no descriptor holds: the access is dropped / returns 0
constexpr int kRows = 1 << 20; // num_records of the ring descriptors (rows); every real row index is far below
// first job of stage s of an NST-stage pipeline (stage NST = end)
constexpr int first_job(int nst, int s)
{
  if (s <= 0)
    return 0;
  if (s >= nst)
    return p2::kJobs;
  if (nst == 2)
    return 12; // 0-11 | 12-19: array 1's two exchange jobs move to the older waves (they win the issue arbitration); a
               // stage must hold both exchange jobs of an array or none (each one's barrier guards the other's window)
  if (nst == 3)
    return s == 1 ? 6 : 13; // 6 full-width | 4 full + 3 half | 7 half-width jobs
  return 5 * s; // four stages of five
}
constexpr int stage_of(int nst, int job)
{
  int s = 0;
  for (int k = 1; k < nst; k++)
    if (job >= first_job(nst, k))
      s = k;
  return s;
}
constexpr int max_jobs(int nst)
{
  int m = 0;
  for (int s = 0; s < nst; s++)
    m = first_job(nst, s + 1) - first_job(nst, s) > m ? first_job(nst, s + 1) - first_job(nst, s) : m;
  return m;
}
constexpr int kWinB = 2 * kBlock * 80; // one exchange window (p2's row order), one per exchange job: 0, 1, 10, 11
constexpr int win_index(int job) { return job == 0 ? 0 : job == 1 ? 1 : job == 10 ? 2 : 3; }
constexpr int kConstsB = 4 * kWinB;
constexpr int kXtB = kConstsB + p2::kJobs * 256;
constexpr int kTilesB = kXtB + p2::kXt * 1024;
constexpr int kFlagB = kTilesB + p2::kJobs * kWsTileFloats * 4; // 256 bytes of counters (below)
// words (ints at kFlagB), one writer each: [4 s + w] barrier generation of wave w of stage s | [16 + 8 q + w] buffers
// handed over by wave w into queue q | [16 + 8 q + 4 + w] buffers taken out by wave w of queue q | [48..51] stage 0's
// command decisions (offset, ready) x 2
constexpr int kQueueB = kFlagB + 256;
// queue q (stage q -> q + 1), wave w, ONE slot: x [64 lanes] f4 | head [64] f4 | cond [64] float | token (4 ints).
// (One slot is enough: a stage hands over at the END of its buffer and the next stage takes at the START of its own.)
constexpr int kSlotB = 64 * 16 + 64 * 16 + 64 * 4 + 16;
constexpr int queue_b(int q, int w) { return kQueueB + (q * 4 + w) * kSlotB; }
// behind the queues: two rows of 64 output samples (by the parity of the command count) — a session that publishes every
// command's completion (A1Args::p_out_host == 2) gathers the last stage's four waves' interleaved frames here, and ONE wave
// stores the row to the host's window with one contiguous write-through instruction
constexpr int out_stage_b(int nst) { return kQueueB + (nst - 1) * 4 * kSlotB; }
constexpr int lds_bytes(int nst) { return out_stage_b(nst) + 2 * 256; }
static_assert(lds_bytes(4) <= 160 * 1024, "p4 LDS layout");
constexpr bool exchange_pairs_ok(int nst)
{
  return stage_of(nst, 0) == stage_of(nst, 1) && stage_of(nst, 10) == stage_of(nst, 11);
}
} // namespace p4

template <int C0, int C1, int ACT_T, bool WT, bool PERSIST, int NST>
__global__ __launch_bounds__(NST * 256) void nam_a1_p4_kernel(const float* __restrict__ blob, const A1Args a)
{
  using namespace mf;
  using il::kOob;
  using il::Ops;
  constexpr int NJ = p2::kJobs, MAXJ = p4::max_jobs(NST);
  static_assert(p4::exchange_pairs_ok(NST), "a stage holds both exchange jobs of an array or none");
  extern __shared__ __attribute__((aligned(16))) float lds_p4[];
  char* const lds = reinterpret_cast<char*>(lds_p4);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wall = uni(tid >> 6);
  const int S = wall >> 2; // stage
  const int w = wall & 3;
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
  const int io_bytes = PERSIST ? 0x7ffffff0 : a.n_frames * 4;
  const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc((void*)(in ? in : st), 0, in ? io_bytes : 0, 0x00020000);
  const auto rsrc_out = __builtin_amdgcn_make_buffer_rsrc((void*)(out ? out : st), 0, out ? io_bytes : 0, 0x00020000);
  using i4 = __attribute__((ext_vector_type(4))) int;
  // rsrc_in once more, as plain dwords (base, stride 0, num_records, flags): for the one load issued from inline asm
  const unsigned long long in_addr = (unsigned long long)(in ? in : st);
  const i4 in_desc = {uni((int)(unsigned)in_addr), uni((int)(unsigned)(in_addr >> 32) & 0xffff), in ? io_bytes : 0, 0x00020000};
  int* wpos_tbl = reinterpret_cast<int*>(st);
  const int wposv = lane < NJ ? wpos_tbl[lane] : 0; // lane r = write position of ring r (ring r = job r) at launch
  int* const flags = reinterpret_cast<int*>(lds + p4::kFlagB);

  // ---- weights -> LDS, once per launch, by every wave (requested first, stored behind the ring requests) ----
  constexpr int NT = NST * 256;
  constexpr int kT4 = (NJ * 256 + NT - 1) / NT; // 16-byte tile records per thread
  f4 tl4[kT4];
  {
    const f4* __restrict__ tsrc = reinterpret_cast<const f4*>(blob + a.tiles_off);
#pragma unroll
    for (int i = 0; i < kT4; i++)
      tl4[i] = tsrc[min(i * NT + tid, NJ * 256 - 1)];
  }
  const f4* __restrict__ csrc = reinterpret_cast<const f4*>(blob + a.consts_off);
  const f4* __restrict__ xsrc = reinterpret_cast<const f4*>(blob + a.xt_off);
  const f4 c0v = csrc[min(tid, NJ * 16 - 1)]; // 320 constant records
  const f4 x0v = xsrc[min(tid, p2::kXt * 64 - 1)]; // 192 extra-tile records

  // Ring descriptors with the row pitch as the stride (64 bytes: 16-channel rows, 32 bytes: 8-channel rows ...): a ring
  // access names its row by INDEX. Lane constants: the lane's channel-quad byte offset inside a row when it appends
  // (app_off) and when it fetches history (fch[array]: the half layout reads quad g & 1), and its frame `t` as an index
  // offset — kNoRow for the lanes of an array that hold no channels (their appends drop out without a compare).
  const unsigned long long st_addr = (unsigned long long)st;
  auto ring_desc = [&](int row_b) {
    return i4{uni((int)(unsigned)st_addr), uni((int)((unsigned)(st_addr >> 32) & 0xffffu) | (row_b << 16)), p4::kRows, 0x00020000};
  };
  const i4 rs_a0 = ring_desc(C0 * 4), rs_a1 = ring_desc(C1 * 4);
  const int fch0 = (int)min(C0 == 8 ? (v_g16 & 16u) : v_g16, (unsigned)(C0 * 4 - 16)); // (8 channels: the half layout)
  const int fch1 = (int)min(C1 == 8 ? (v_g16 & 16u) : v_g16, (unsigned)(C1 * 4 - 16));
  auto tl_app_of = [&](int nv, int C) { return (t < nv && (int)v_g16 <= 16 * (C / 4 - 1)) ? t : p4::kNoRow; };
  int tl_app0 = tl_app_of(kBlock, C0), tl_app1 = tl_app_of(kBlock, C1);
  // the write positions of this stage's rings as SCALARS (wp[u] = ring of job J0 + u): advanced on the scalar unit
  int wp[MAXJ];
  il::for_each_index(
    [&](auto s_tag) {
      constexpr int SS = decltype(s_tag)::value;
      if (S == SS)
      {
        constexpr int J0 = p4::first_job(NST, SS), NJS = p4::first_job(NST, SS + 1) - J0;
#pragma unroll
        for (int u = 0; u < MAXJ; u++)
          wp[u] = u < NJS ? __builtin_amdgcn_readlane(wposv, J0 + (u < NJS ? u : 0)) : 0;
      }
    },
    std::make_integer_sequence<int, NST>{});
  // the ring requests of job TJ for the block AHEAD blocks after the one the write positions stand at; `wpj`: the
  // job's write position, `valid`: wave-uniform
  auto fetch = [&](f4& ra, f4& rb, auto tj_tag, auto ahead_tag, bool valid, int tl, int wpj) {
    constexpr int TJ = decltype(tj_tag)::value;
    constexpr int JF = (TJ + NJ - p2::kDepth) % NJ; // table position whose entry describes job TJ
    constexpr int AHEAD = decltype(ahead_tag)::value;
    constexpr IlFetch F = p2::fetch(C0, C1, JF);
    constexpr bool arr1 = TJ >= p2::kLayers;
#pragma unroll
    for (int q = 0; q < 2; q++)
    {
      constexpr int LA = F.LA, LB = F.LB, nA = F.nA, nB = F.nB;
      const int L = q == 0 ? LA : LB;
      const int n = q == 0 ? nA : nB;
      if (L > 0) // (compile time: exchange jobs have one request)
      {
        // row of lane frame tl: (wpj + AHEAD * 64 - L + tl) mod R — the lane-independent part on the scalar unit
        int sb_ = wpj + (AHEAD ? kBlock : 0) - L;
        sb_ += sb_ < 0 ? F.R : 0;
        sb_ -= sb_ >= F.R ? F.R : 0;
        sb_ = valid ? sb_ : p4::kNoRow;
        const int tq = n >= 16 ? tl : (tl < 4 * n ? tl : p4::kNoRow); // lanes j < n only
        const unsigned v = (unsigned)(sb_ + tq);
        const int idx = (int)min(v, v - (unsigned)F.R);
        const f4 r = p4_sb_load(arr1 ? rs_a1 : rs_a0, idx, arr1 ? fch1 : fch0, F.ring_b, 0);
        if (q == 0)
          ra = r;
        else
          rb = r;
      }
    }
  };
  // Request slots: one per job of the wave's OWN stage, refilled for the next buffer as soon as the job has consumed it
  f4 sa[MAXJ], sb[MAXJ];
  float inp = 0.0f;

  // PERSIST + WT: a session whose output window is HOST memory (the blocking host entry points, A1Args::p_out_host): the
  // results are plain stores and one release fence per workgroup publishes them when the launch leaves. (Ring appends stay
  // write-back as in every session: written through as well, the buffer takes 44 us instead of 37 at 256 streams,
  // profiles/r03/persist_io_store_scope_variants.txt.) WT without PERSIST: a short launch, ring appends written through.
  constexpr bool kOutHost = PERSIST && WT;
  constexpr int kInAux = PERSIST ? 17 : 0; // session inputs bypass the caches (the caller may rewrite the buffer between commands)
  auto ring_load = [&](unsigned s_) {
    return __hip_atomic_load(a.p_ring + (s_ & (unsigned)a.p_ring_mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  };
  // Synchronisation words in LDS, each with ONE writer (so plain stores, no atomics): wave w of stage s publishes its
  // barrier generation in flags[4 s + w]; producer wave w of queue q its count of buffers handed over in
  // flags[16 + 8 q + w], the consumer wave its count taken out in flags[16 + 8 q + 4 + w]. The waiting side polls
  // from inline asm — a wait loop in C++ is a LOOP to the compiler: it splits the live ranges of the request slots
  // around it, shuffles them with copies at the joins and drains every ring request in flight (vmcnt(0)) to do so; an
  // asm statement is straight-line code. One wave's LDS operations execute in program order: "data, then flag" needs
  // nothing in between. The queue's data moves inside the same asm statements as its flags (volatile asm statements
  // keep their order; plain LDS accesses may move across them, which is harmless: different addresses).
  const unsigned flag_b = (unsigned)p4::kFlagB;
  auto wait_word = [&](unsigned byte_addr, int want) { // until the word has reached `want`
    int tmp;
    asm volatile("1:\n\tds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_sub_u32 %0, %0, %2\n\tv_cmp_gt_i32 vcc, 0, %0\n\t"
                 "s_cbranch_vccz 2f\n\ts_sleep 1\n\ts_branch 1b\n2:"
                 : "=&v"(tmp)
                 : "v"(byte_addr), "v"(want)
                 : "vcc");
  };
  int bar_gen = 0; // this wave's count of stage barriers passed
  auto stage_barrier = [&]() { // the four waves of this stage: every wave publishes its generation, waits for all four
    asm volatile("" ::: "memory"); // (the exchange window's plain stores / loads stay on their side)
    bar_gen++;
    const unsigned mine = flag_b + (unsigned)(4 * S + w) * 4u, all4 = flag_b + (unsigned)(4 * S) * 4u;
    int t0, t1, t2, t3;
    asm volatile("ds_write_b32 %4, %5\n"
                 "1:\n\tds_read_b32 %0, %6\n\tds_read_b32 %1, %6 offset:4\n\tds_read_b32 %2, %6 offset:8\n\tds_read_b32 %3, %6 offset:12\n\t"
                 "s_waitcnt lgkmcnt(0)\n\tv_min_i32 %0, %0, %1\n\tv_min_i32 %2, %2, %3\n\tv_min_i32 %0, %0, %2\n\t"
                 "v_sub_u32 %0, %0, %5\n\tv_cmp_gt_i32 vcc, 0, %0\n\ts_cbranch_vccz 2f\n\ts_sleep 1\n\ts_branch 1b\n2:"
                 : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                 : "v"(mine), "v"(bar_gen), "v"(all4)
                 : "vcc");
    asm volatile("" ::: "memory");
  };
  // queue q, wave w: slot = x [64 lanes] f4 | head [64] f4 | cond [64] float | token (4 ints: byte offset of the buffer,
  // valid frames, EXIT, "another buffer follows")
  auto queue_put = [&](int q, int k, const f4& vx, const f4& vh, float vc, const i4& tok) {
    const unsigned slot = (unsigned)p4::kQueueB + (unsigned)((q * 4 + w) * p4::kSlotB);
    const unsigned prod = flag_b + (unsigned)(16 + 8 * q + w) * 4u, cons = flag_b + (unsigned)(16 + 8 * q + 4 + w) * 4u;
    wait_word(cons, k); // the slot is free once buffer k - 1 has been taken out of it
    // plain LDS stores (the compiler pads the job's last MFMA and the store of its result with the wait states the
    // hardware requires — it does not do that for the text of an asm statement: an asm version of this block read x
    // before the matrix pipe had written it), pinned in place on both sides
    asm volatile("" ::: "memory");
    lds_st4(lds, slot + (unsigned)lane * 16u, vx);
    lds_st4(lds, slot + 1024u + (unsigned)lane * 16u, vh);
    *reinterpret_cast<float*>(lds + slot + 2048u + (unsigned)lane * 4u) = vc;
    if (lane == 0)
      *reinterpret_cast<i4*>(lds + slot + 2304u) = tok;
    asm volatile("" ::: "memory");
    if (lane == 0)
      __hip_atomic_store(reinterpret_cast<int*>(lds + prod), k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
  };
  auto queue_take = [&](int q, int k, f4& vx, f4& vh, float& vc, i4& tok) {
    const unsigned slot = (unsigned)p4::kQueueB + (unsigned)((q * 4 + w) * p4::kSlotB);
    const unsigned prod = flag_b + (unsigned)(16 + 8 * q + w) * 4u, cons = flag_b + (unsigned)(16 + 8 * q + 4 + w) * 4u;
    wait_word(prod, k + 1);
    asm volatile("" ::: "memory");
    tok = *reinterpret_cast<const i4*>(lds + slot + 2304u);
    vx = lds_ld4(lds, slot + (unsigned)lane * 16u);
    vh = lds_ld4(lds, slot + 1024u + (unsigned)lane * 16u);
    vc = *reinterpret_cast<const float*>(lds + slot + 2048u + (unsigned)lane * 4u);
    asm volatile("" ::"v"(vx), "v"(vh), "v"(vc), "v"(tok) : "memory"); // (the values are in registers: the slot may be reused)
    if (lane == 0)
      __hip_atomic_store(reinterpret_cast<int*>(lds + cons), k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
  };

  // every ring request of the first buffer (they depend on the state only, not on the command): each stage its own jobs
  il::for_each_index(
    [&](auto s_tag) {
      constexpr int SS = decltype(s_tag)::value;
      if (S == SS)
      {
        constexpr int J0 = p4::first_job(NST, SS), NJS = p4::first_job(NST, SS + 1) - J0;
        il::for_each_index(
          [&](auto u_tag) {
            constexpr int U = decltype(u_tag)::value;
            fetch(sa[U], sb[U], std::integral_constant<int, J0 + U>{}, std::integral_constant<int, 0>{}, true, t, wp[U]);
          },
          std::make_integer_sequence<int, NJS>{});
      }
    },
    std::make_integer_sequence<int, NST>{});

  unsigned na = 0; // PERSIST, stage 0: commands finished by this stage (its current command carries tag na + 1)
  unsigned done = 0; // PERSIST: commands consumed before this launch (+ finished by the last stage during it)
  unsigned boff0 = 0; // stage 0: byte offset of its first buffer
  if (tid < 64)
    flags[tid] = 0;
  if constexpr (PERSIST)
  {
    const bool by_value = a.p_seq0 >= 0;
    done = na = by_value ? (unsigned)a.p_seq0 : a.p_cons[blockIdx.x];
    bool ready = true;
    unsigned lo = (unsigned)a.p_cmd0;
    if (!by_value)
    {
      lds_barrier(); // (the counters are zero)
      if (wall == 0)
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
          flags[48] = (int)(unsigned)v;
          flags[49] = (unsigned)(v >> 32) == na + 1 ? 1 : 0;
        }
      }
      lds_barrier();
      ready = uni(flags[49]) != 0;
      lo = (unsigned)uni(flags[48]);
      lds_barrier();
    }
    if (!ready)
    {
      // nothing to do (the doorbell this launch was started for has been consumed by its predecessor)
      if (wall == 0 && lane == 0)
      {
        a.p_cons[blockIdx.x] = done;
        __hip_atomic_store(a.p_done + blockIdx.x, done | 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      return;
    }
    boff0 = lo * 4u;
  }
  if (S == 0)
    inp = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_in, t * 4, uni((int)boff0), kInAux));
  // the weights (requested before the ring rows, so they are here first)
#pragma unroll
  for (int i = 0; i < kT4; i++)
    if (i * NT + tid < NJ * 256)
      lds_st4(lds, (unsigned)p4::kTilesB + (unsigned)(i * NT + tid) * 16u, tl4[i]);
  if (tid < NJ * 16)
    lds_st4(lds, (unsigned)p4::kConstsB + (unsigned)tid * 16u, c0v);
  if (tid < p2::kXt * 64)
    lds_st4(lds, (unsigned)p4::kXtB + (unsigned)tid * 16u, x0v);
  lds_barrier();

  auto load_ops = [&](Ops& o, auto j_tag) {
    constexpr int JN = decltype(j_tag)::value; // the job whose operands are read
    constexpr unsigned consts_b = p4::kConstsB + JN * 256, tiles_b = p4::kTilesB + JN * 4096;
#pragma unroll
    for (int q = 0; q < 4; q++)
      o.t[q] = lds_ld4(lds, v_lane16 + tiles_b + 1024u * q);
    o.bv = lds_ld4(lds, v_g16 + consts_b);
    o.mv = lds_ld4(lds, v_g16 + consts_b + 64u);
    o.b1v = lds_ld4(lds, v_g16 + consts_b + 128u);
  };
  auto load_extra = [&](f4& xt, f4& ev, auto j_tag) {
    constexpr int JN = decltype(j_tag)::value;
    xt = lds_ld4(lds, v_lane16 + (unsigned)(p4::kXtB + p2::xt_index(JN) * 1024));
    ev = lds_ld4(lds, v_g16 + (unsigned)(p4::kConstsB + JN * 256) + 192u);
  };

  f4 x = {0.f, 0.f, 0.f, 0.f}, head = {0.f, 0.f, 0.f, 0.f};
  int nvalid = kBlock; // frames of the buffer this wave's stage is working on
  float cond = 0.0f;
  Ops O;
  unsigned long long spec_cmd = 0; // PERSIST, stage 0: this wave's early look at the next command ...
  float inp_spec = 0.0f; // ... and the input sample it requested on a hit
  bool more = false; // this wave's stage has another buffer behind the current one (its slots are refilled for it)
  int blk = 0; // buffers this wave's stage has finished
  unsigned boff = 0; // byte offset of the current buffer in the stream's row

  // One job, everything about it known at compile time. With three wavefronts on a SIMD the INSTRUCTION COUNT is the
  // time again (profiles/r03: the vector ALU's issue slots and the matrix pipe are each ~43 % busy and do not overlap
  // further whatever the wave count), not a lone wave's latency chain — so, unlike nam_a1_p2_kernel's job: ONE
  // accumulator chain per matrix product (bias and input mixin seed it; no partial sums to add up afterwards, no tap
  // products computed a job early), ring rows addressed by index (no address arithmetic beyond the wrap), write
  // positions on the scalar unit. model.cpp:183-393 for the plain layer: z = act(conv(x) + mixin(cond));
  // head += z; x += layer1x1(z).
  auto job = [&](auto j_tag) {
    constexpr int JI = decltype(j_tag)::value;
    constexpr int SJ = p4::stage_of(NST, JI);
    constexpr int J0 = p4::first_job(NST, SJ);
    constexpr int SL = JI - J0; // the request slot
    constexpr IlDesc J = p2::desc(C0, C1, 0, JI);
    constexpr int flags_j = J.flags;
    constexpr int NK = (flags_j & CD_HALF) ? 2 : 4;
    constexpr unsigned g16max = (unsigned)J.gp;
    constexpr bool arr1 = JI >= p2::kLayers;
    const int act = a.act; // (only read by the run-time-dispatch instantiation)
    __builtin_amdgcn_sched_barrier(0);
    int tl = t;
    unsigned gl16 = v_g16;
    asm volatile("" : "+v"(tl), "+v"(gl16));
    load_ops(O, j_tag);
    f4 xt = {0.f, 0.f, 0.f, 0.f}, ev = {0.f, 0.f, 0.f, 0.f};
    if constexpr ((flags_j & (CD_X0 | CD_PRE_HEAD | CD_POST_RECH | CD_POST_OUT)) != 0)
      load_extra(xt, ev, j_tag);
    const f4 Sa = sa[SL], Sb = sb[SL];
    if constexpr (J.kind == IL_EXCH)
      asm volatile("" ::"v"(Sa));
    else
      asm volatile("" ::"v"(Sa), "v"(Sb)); // one wait for the whole slot (the oldest requests in flight)
    if constexpr ((flags_j & CD_X0) != 0)
    {
      cond = inp; // this buffer's input sample (requested a buffer ago)
      if constexpr (!PERSIST) // next block's (offset beyond the launch's frames -> 0)
        inp = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_in, tl * 4, uni((blk + 1) * (kBlock * 4)), 0));
      x = ev * cond;
      head = f4{0.f, 0.f, 0.f, 0.f};
    }
    // this job's input -> its history ring: row (position + frame) mod R; lanes without channels carry kNoRow
    const int wpj = wp[SL];
    {
      const unsigned v = (unsigned)(wpj + (arr1 ? tl_app1 : tl_app0));
      const int widx = (int)min(v, v - (unsigned)J.R);
      p4_sb_store(x, arr1 ? rs_a1 : rs_a0, widx, (int)gl16, J.ring_b, WT && !PERSIST ? 17 : 0);
    }
    // persistent session, stage 0: every wave looks at the next ring slot in job 1 and, when the command is already
    // there, requests the next buffer's input sample from it in job 3 (unconditional load, out-of-range offset on a miss)
    if constexpr (PERSIST && JI == 1)
      spec_cmd = ring_load(na + 1);
    if constexpr (PERSIST && JI == 3)
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
      // exchange between the four waves of this stage: one window per exchange job, row order as in p2 (each
      // window's next use sits behind the stage barrier of the array's other exchange job)
      constexpr unsigned kRowB = 80u;
      auto win_off = [&](unsigned F) { return ((F & 64u) + ((F & 3u) << 4) + ((F & 63u) >> 2)) * kRowB; };
      constexpr unsigned wb = (unsigned)p4::win_index(JI) * (unsigned)p4::kWinB;
      if (gl16 <= g16max)
      {
        lds_st4(lds, wb + win_off((unsigned)(kBlock + tl)) + gl16, x);
        lds_st4(lds, wb + win_off((unsigned)tl) + gl16, Sa);
      }
      stage_barrier();
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
    if constexpr ((flags_j & CD_PRE_HEAD) != 0)
      head = ((flags_j & CD_PREV_HALF) ? mfma_n<2>(xt, head, ev) : mfma_n<4>(xt, head, ev));
    // conv + mixin: one chain seeded with bias + mixin * condition
    f4 acc = __builtin_elementwise_fma(O.mv, f4{cond, cond, cond, cond}, O.bv);
#pragma unroll
    for (int s = 0; s < NK; s++)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(O.t[0][s], bt0[s], acc, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < NK; s++)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(O.t[1][s], bt1[s], acc, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < NK; s++)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(O.t[2][s], x[s], acc, 0, 0, 0);
    // The same job of this stage's NEXT buffer: its requests go into the slot just consumed — issued HERE, behind the
    // last use of the slot's old rows (the tap operands of the chain above): the new rows can then land in the very same
    // registers. Requested while the old rows are still live they need other registers, the two sets swap roles every
    // buffer, and the compiler pays for that with copies at the loop's back edge — behind a wait for the loads.
    fetch(sa[SL], sb[SL], j_tag, std::integral_constant<int, 1>{}, more, tl, wpj);
    const f4 z = act4<ACT_T>(act, NK == 2 ? f4{acc[0], acc[1], acc[0], acc[1]} : acc, act_p0);
    head += z;
    asm volatile("" : "+v"(head)); // (pin the accumulator: kernel_a1_p2.hip)
    // layer 1x1 + residual: one chain seeded with x + bias
    f4 y = x + O.b1v;
#pragma unroll
    for (int s = 0; s < NK; s++)
      y = __builtin_amdgcn_mfma_f32_16x16x4f32(O.t[3][s], z[s], y, 0, 0, 0);
    x = y;
    if constexpr ((flags_j & CD_POST_OUT) != 0)
    {
      const float yout = head_scale * mfma_n<NK>(xt, head, ev)[0];
      const bool ok = gl16 == 0 && tl < nvalid;
      // a session's results: written through to DEVICE memory (sc0 sc1: whoever reads them next may not be ordered
      // behind this launch's end). To HOST memory such stores complete one at a time (~1 us each, 1 ms per buffer at
      // 256 streams: profiles/r03/persist_io_store_scope_variants.txt): plain stores there, made visible by the release
      // fence before the completion word (kOutHost below)
      if (kOutHost && a.p_out_host == 2)
      {
        // every command a completion of its own (the stage loop below): the wave's sixteen frames — every fourth sample of the
        // row — go to the staging row in LDS; written through to HOST memory from here they would be sixteen 4-byte PCIe
        // writes per wave (measured: 1.75 ms per 64-frame call at 256 streams)
        if (ok)
          *reinterpret_cast<float*>(lds + p4::out_stage_b(NST) + (int)(done & 1u) * 256 + tl * 4) = yout;
      }
      else
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yout), rsrc_out, ok ? tl * 4 : (int)kOob, uni((int)boff),
                                              PERSIST && !kOutHost ? 17 : 0);
    }
    else if constexpr ((flags_j & CD_POST_RECH) != 0)
      x = mfma_n<NK>(xt, x, f4{0.f, 0.f, 0.f, 0.f});
    // this ring moves on by the buffer's frames (scalar unit)
    {
      int np = wpj + nvalid;
      np -= np >= J.R ? J.R : 0;
      wp[SL] = np;
    }
  };

  // ---- the stage loops ----
  // queue q = stage q -> stage q + 1, per wave two slots: data + token {byte offset of the buffer, valid frames, 1 = EXIT}
  auto run = [&](auto s_tag) {
    constexpr int SS = decltype(s_tag)::value;
    constexpr int J0 = p4::first_job(NST, SS), NJS = p4::first_job(NST, SS + 1) - J0;
    constexpr bool FIRST = SS == 0, LAST = SS == NST - 1;
    constexpr int QIN = SS - 1, QOUT = SS;
    boff = boff0;
    bool have = !FIRST || n_blocks > 0; // FIRST: does this stage have a buffer to start (later stages find out from the token)
#pragma unroll 1
    for (int k = 0;; k++)
    {
      bool exit_tok = false;
      if constexpr (FIRST)
      {
        exit_tok = !have;
        nvalid = PERSIST ? kBlock : min(kBlock, a.n_frames - k * kBlock);
        more = PERSIST || k + 1 < n_blocks;
      }
      else
      {
        // the token and the hand-over of buffer k from the same wave of the previous stage
        i4 tok;
        queue_take(QIN, k, x, head, cond, tok);
        boff = (unsigned)uni(tok[0]);
        nvalid = uni(tok[1]);
        exit_tok = uni(tok[2]) != 0;
        more = PERSIST || uni(tok[3]) != 0;
      }
      // hand-over to the same wave of the next stage
      auto hand_over = [&](bool is_exit) {
        queue_put(QOUT < 0 ? 0 : QOUT, k, x, head, cond, i4{(int)boff, nvalid, is_exit ? 1 : 0, more ? 1 : 0});
      };
      if (exit_tok)
      {
        // nothing (more) to do: pass the EXIT token on and leave. (In front of the jobs and out of the loop — with the
        // jobs on a conditional path inside it, the request slots would merge with a "skipped" copy at the join and the
        // compiler would drain every request in flight, vmcnt(0), to shuffle them at the end of each buffer.)
        if constexpr (!LAST)
          hand_over(true);
        break;
      }
      if constexpr (!PERSIST)
      {
        if (nvalid != kBlock) // a ragged last block: only its frames are appended
        {
          tl_app0 = tl_app_of(nvalid, C0);
          tl_app1 = tl_app_of(nvalid, C1);
        }
      }
      il::for_each_index([&](auto u_tag) { job(std::integral_constant<int, J0 + decltype(u_tag)::value>{}); },
                         std::make_integer_sequence<int, NJS>{});
      blk++;
      if constexpr (!LAST)
        hand_over(false);
      else if constexpr (PERSIST)
      {
        done++;
        if (kOutHost && a.p_out_host == 2)
        {
          // a blocking host caller waits for THIS command (round 6: nam_hip_batch_process_f32 back to back rides one lingering
          // launch, kernel_a1_q.hip's ticket protocol): the stage's four waves meet, wave 0 stores the row (one contiguous
          // instruction, system scope, written through), has it acknowledged and counts the workgroup in; the last workgroup
          // through the command stores the host's completion word (kernels.h: p_cmd_count / p_cmd_done)
          stage_barrier(); // (a wave's LDS operations execute in order: its samples are in the row before its word says so)
          if (w == 0)
          {
            const float yrow = *reinterpret_cast<const float*>(lds + p4::out_stage_b(NST) + (int)((done - 1u) & 1u) * 256 + lane * 4);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yrow), rsrc_out, lane < nvalid ? lane * 4 : (int)kOob, uni((int)boff), 17);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // acknowledged (system scope, written through)
            unsigned before = 0u;
            const unsigned cslot = (done - 1u) & (unsigned)a.p_ring_mask;
            if (lane == 0)
              before = __hip_atomic_fetch_add(a.p_cmd_count + cslot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)uni((int)before) == gridDim.x - 1u && lane == 0)
            {
              __hip_atomic_store(a.p_cmd_count + cslot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(a.p_cmd_done + cslot, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
              __hip_atomic_fetch_max(a.p_cmd_count + a.p_ring_mask + 1, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (il_common.h: session_wait_command)
            }
          }
        }
        if (w == 0 && lane == 0 && (done & 15u) == 0u) // progress for the host's ring bookkeeping (not a completion signal)
          __hip_atomic_store(a.p_prog + blockIdx.x, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      if constexpr (FIRST)
      {
        // the next buffer of this stage
        if constexpr (PERSIST)
        {
          if (w == 0)
          {
            const unsigned tag = na + 2u; // the command behind the one just finished
            unsigned long long v = spec_cmd;
            if ((unsigned)(v >> 32) != tag)
            {
              // the early look missed: look again; while the later stages still work, for a few microseconds (bounded: the
              // launch never waits for a command) — or for A1Args::p_linger when the session serves a host caller that hands
              // one buffer in after the other (il_common.h: session_wait_command, the rules of a lingering launch)
              v = session_wait_command<16>(a, ring_load, tag, ring_load(tag - 1u), (long long)(a.p_linger > 0 ? a.p_linger : 100)); // (1 us unless the launch lingers)
            }
            if (lane == 0)
            {
              flags[48 + 2 * (k & 1)] = (int)(unsigned)v;
              flags[48 + 2 * (k & 1) + 1] = (unsigned)(v >> 32) == tag ? 1 : 0;
              if ((unsigned)(v >> 32) != tag)
                session_leaving(a); // (the others stop waiting for workgroups behind them)
            }
          }
          stage_barrier(); // wave 0's decision is the stage's (two decision slots by parity: the one written now was
                           // read by every wave before it passed the previous buffer's barrier)
          have = uni(flags[48 + 2 * (k & 1) + 1]) != 0;
          const unsigned next_off = (unsigned)uni(flags[48 + 2 * (k & 1)]) * 4u;
          na++;
          if (have)
          {
            boff = next_off;
            const bool mine = (unsigned)(spec_cmd >> 32) == na + 1u && (unsigned)spec_cmd * 4u == next_off;
            inp = inp_spec;
            if (!mine)
            {
              // (rare: this wave's look came too early) load now and wait right here, inside the asm — a load the
              // compiler can see would be the youngest operation in flight when job 0 consumes it, and its wait-count
              // pass would drain every ring request in front of job 0 on every round
              const int voff = t * 4, soff = uni((int)next_off);
              const i4 rs = in_desc;
              asm volatile("buffer_load_dword %0, %1, %2, %3 offen sc0 sc1\n\ts_waitcnt vmcnt(0)"
                           : "=v"(inp)
                           : "v"(voff), "s"(rs), "s"(soff)
                           : "memory");
            }
          }
        }
        else
        {
          have = k + 1 < n_blocks;
          boff = (unsigned)(k + 1) * (kBlock * 4u);
        }
      }
    }
  };
  il::for_each_index(
    [&](auto s_tag) {
      if (S == decltype(s_tag)::value)
        run(s_tag);
    },
    std::make_integer_sequence<int, NST>{});

  // the write positions of this stage's rings go back into the state (lane r of the stage's wave 0 = ring r)
  il::for_each_index(
    [&](auto s_tag) {
      constexpr int SS = decltype(s_tag)::value;
      if (S == SS && w == 0)
      {
        constexpr int J0 = p4::first_job(NST, SS), NJS = p4::first_job(NST, SS + 1) - J0;
        int v = 0;
#pragma unroll
        for (int u = 0; u < NJS; u++)
          v = lane == J0 + u ? wp[u] : v;
        if (lane >= J0 && lane < J0 + NJS)
          wpos_tbl[lane] = v;
      }
    },
    std::make_integer_sequence<int, NST>{});
  if constexpr (PERSIST)
  {
    // results visible (the session's outputs are stored write-through), then the consumed-command count: the last
    // stage's wave 0 knows it
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    if constexpr (kOutHost)
    {
      // system scope: the plain stores to host memory are out. ONE wave per workgroup: the write-back behind the fence is
      // the whole L2's, not a wave's, and every wave's stores reached L2 in front of the barrier (twelve waves x 256
      // workgroups each asking for it: +45 us per buffer)
      if (S == NST - 1 && w == 0)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    }
    if (S == NST - 1 && w == 0 && lane == 0)
    {
      a.p_cons[blockIdx.x] = done;
      __hip_atomic_store(a.p_done + blockIdx.x, done | 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

namespace
{
#ifndef NAM_P4_STAGES
#define NAM_P4_STAGES 3
#endif
constexpr int kP4Stages = NAM_P4_STAGES;

// the instantiation's code object on the device and its dynamic-LDS limit raised (what the first launch would otherwise pay: ~1.6 ms)
template <int C0, int C1, int ACT_T, bool WT, bool PERSIST>
hipError_t p4_ready()
{
  static DynamicLdsLimit lds_limit; // per instantiation, tracked per device (kernels.h)
  return lds_limit.ensure(reinterpret_cast<const void*>(&nam_a1_p4_kernel<C0, C1, ACT_T, WT, PERSIST, kP4Stages>), p4::lds_bytes(kP4Stages));
}
template <int C0, int C1, int ACT_T, bool WT, bool PERSIST = false>
hipError_t launch_p4_inst(const A1Args& a, int n_blocks, hipStream_t stream)
{
  constexpr int lds_bytes = p4::lds_bytes(kP4Stages);
  const hipError_t e = p4_ready<C0, C1, ACT_T, WT, PERSIST>();
  if (e != hipSuccess)
    return e;
  nam_launch((nam_a1_p4_kernel<C0, C1, ACT_T, WT, PERSIST, kP4Stages>), dim3(n_blocks), dim3(kP4Stages * 256), lds_bytes, stream,
                     a.blob, a);
  return hipGetLastError();
}
template <int C0, int C1>
hipError_t launch_p4_shape(const A1Args& a, int n_blocks, int act, hipStream_t stream)
{
  if (a.p_ring) // persistent session: write-back ring appends (kernel_a1_p2.hip: launch_p2_shape) ...
  {
    const bool oh = a.p_out_host != 0; // ... unless its results go to host memory (kOutHost)
    if (act == ACT_FASTTANH)
      return oh ? launch_p4_inst<C0, C1, ACT_FASTTANH, true, true>(a, n_blocks, stream) : launch_p4_inst<C0, C1, ACT_FASTTANH, false, true>(a, n_blocks, stream);
    if (act == ACT_TANH)
      return oh ? launch_p4_inst<C0, C1, ACT_TANH, true, true>(a, n_blocks, stream) : launch_p4_inst<C0, C1, ACT_TANH, false, true>(a, n_blocks, stream);
    return oh ? launch_p4_inst<C0, C1, -1, true, true>(a, n_blocks, stream) : launch_p4_inst<C0, C1, -1, false, true>(a, n_blocks, stream);
  }
  const bool wt = a.n_frames <= 2 * kBlock; // short launches write ring appends through (device_common.h: ring_store)
  if (act == ACT_FASTTANH)
    return wt ? launch_p4_inst<C0, C1, ACT_FASTTANH, true>(a, n_blocks, stream) : launch_p4_inst<C0, C1, ACT_FASTTANH, false>(a, n_blocks, stream);
  if (act == ACT_TANH)
    return wt ? launch_p4_inst<C0, C1, ACT_TANH, true>(a, n_blocks, stream) : launch_p4_inst<C0, C1, ACT_TANH, false>(a, n_blocks, stream);
  return wt ? launch_p4_inst<C0, C1, -1, true>(a, n_blocks, stream) : launch_p4_inst<C0, C1, -1, false>(a, n_blocks, stream);
}
} // namespace

// A session of the 16 / 8 topology may switch to this kernel in the middle of a caller's real-time loop (api_launch.cpp:
// PersistSession::short_bursts): its session instantiation is made ready when the session starts, not at the switch.
hipError_t preload_a1_p4_session(int c0, int c1, int act, bool out_host)
{
  if (c0 != 16 || c1 != 8)
    return hipSuccess;
  if (act == ACT_FASTTANH)
    return out_host ? p4_ready<16, 8, ACT_FASTTANH, true, true>() : p4_ready<16, 8, ACT_FASTTANH, false, true>();
  if (act == ACT_TANH)
    return out_host ? p4_ready<16, 8, ACT_TANH, true, true>() : p4_ready<16, 8, ACT_TANH, false, true>();
  return hipSuccess;
}

hipError_t launch_a1_p4(const A1Args& a, int n_blocks, int c0, int c1, int act, hipStream_t stream)
{
  if (c0 == 16 && c1 == 8)
    return launch_p4_shape<16, 8>(a, n_blocks, act, stream);
  if (c0 == 12 && c1 == 8)
    return launch_p4_shape<12, 8>(a, n_blocks, act, stream);
  if (c0 == 8 && c1 == 4)
    return launch_p4_shape<8, 4>(a, n_blocks, act, stream);
  return hipErrorInvalidValue;
}

} // namespace namhip
