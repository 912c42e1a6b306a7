// kernel_a1_q.hip — nam_a1_q_kernel: the official A1 "standard" topology (aq_table.h) as a pipeline of SIXTEEN ONE-WAVE STAGES
// whose hand-over medium is the next layer's history ring itself, most rings resident in LDS for the whole launch.
#include "device_common.h"
#include "il_common.h"
#include "aq_table.h"

#ifndef NAM_AQ_SLEEP
#define NAM_AQ_SLEEP 1 // s_sleep between two looks at a hand-over word (x 64 cycles); 0 and 2 measured level (profiles/r05/a1q_variants.txt)
#endif
#define NAM_AQ_STR2(x) #x
#define NAM_AQ_STR(x) NAM_AQ_STR2(x)

namespace namhip
{

// ================================================================================================
// What nam_a1_p4_kernel (kernel_a1_p4.hip; read its header first) leaves on the table, from its counters
// (profiles/r03/counters_c2_p4_resident.json): the 8-channel array pays for 16 x 16 x 4 tiles that are half padding and
// computes its activation on duplicated rows (7.9 k matrix + 5.7 k vector cycles per SIMD and buffer, and an fp32 MFMA
// never overlaps a vector instruction on this chip); every layer's ring row goes to HBM and comes back (33 MB per step at
// 256 streams: 5.3 us at the achievable HBM rate — the floor under ANY instruction tuning); every wave reads every weight
// tile from LDS for its 16 frames. This kernel changes the shape of the work (model.cpp:183-393 / 463-549 unchanged):
//   * one WAVE per stage and 64 frames per wave. Array 0 (16 channels) stays on v_mfma_f32_16x16x4_f32 — no padding there —
//     as FOUR frame groups per wave: lane (g = lane / 16, n = lane % 16) holds channels 4 g .. 4 g + 3 of frames 16 F + n,
//     F = 0 .. 3; D layout = B layout (kernel_a1_p2.hip's trick), four independent accumulator chains per product, and the
//     A operands — 4 registers per 16 x 16 matrix — stay IN REGISTERS for the whole launch (16 per layer): no weight
//     traffic at all. Array 1 (8 channels), the transition and the head run one lane per frame on
//     v_mfma_f32_4x4x1_16b_f32 (kernel_kq.hip): no padded rows, half the activations;
//   * stage -> stage: the producer writes its output rows STRAIGHT INTO THE CONSUMER'S FIRST RING (planes of 16-byte rows
//     in LDS) and the head accumulator / input sample / token into a one-slot queue; "produced" / "consumed" words with one
//     writer each. The consumer signals "consumed" right behind the LDS reads of its first job's taps (LDS runs a wave's
//     accesses in order: the producer's next rows cannot overtake them);
//   * rings with (K - 1) d + 64 rows up to 12 KB (array 0: d <= 64, array 1: d <= 128) live in LDS from the first buffer
//     of a launch to its last: loaded from the stream state when the launch starts, written back (same R, same write
//     position: the state stays the one every A1 kernel shares) when it leaves. Five rings stay in HBM (array 0: d = 128,
//     256, 512; array 1: d = 256, 512): appended every buffer, their taps — all at least two buffers old — requested one
//     buffer ahead into registers. HBM traffic per stream and buffer: 48 KB instead of 114.
// Sums: one chain per product seeded with bias + mixin * input, taps oldest first (as nam_a1_p4_kernel / nam_kq_kernel).
// ================================================================================================
using aq_i4 = __attribute__((ext_vector_type(4))) int;
__device__ mf::f4 aq_sb_load4(aq_i4 rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.v4f32");
__device__ void aq_sb_store4(mf::f4 v, aq_i4 rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.store.v4f32");

// The activations with every fused multiply-add spelled out and contraction off: the compiler's own choice of what to fuse
// differs between instantiations of one kernel (it depends on the code around the expression), and a session must render
// bit for bit what a plain launch renders (tests/test_gpu_breadth.py: test_pipelined_kernel_against_the_four_wave_kernel).
// Fasttanh: NAM/activations.h:91-98 (the rational, rearranged: see below); Tanh: 1 - 2 / (exp(2 x) + 1).
#pragma clang fp contract(off)
template <int ACT_T>
__device__ __forceinline__ mf::f4 aq_act4(const mf::f4& v)
{
  mf::f4 r;
#pragma unroll
  for (int i = 0; i < 4; i++)
  {
    const float x = v[i];
    if constexpr (ACT_T == ACT_FASTTANH)
    {
      // x (a + a |x| + (b + c |x|) x^2) / (d + (d + x^2) |x + e x |x||) in ten instructions instead of eleven: |x + e x |x|| =
      // |x| (1 + e |x|) (e > 0), and with q = x^2 + d the numerator / x is t1 q + (t2 - d t1), whose second factor is linear
      // in |x| like t2 — x^2 never exists by itself (kernel_lstm.hip: lrow::ratio)
      constexpr float kA1 = (float)(2.45550750702956 - 2.44506634652299 * 0.821226666969744), kA0 = (float)(2.45550750702956 - 2.44506634652299 * 0.893229853513558);
      const float ax = __builtin_fabsf(x);
      const float q = __builtin_fmaf(ax, ax, 2.44506634652299f);
      const float w = __builtin_fmaf(0.814642734961073f, ax, 1.0f);
      const float t1 = __builtin_fmaf(0.821226666969744f, ax, 0.893229853513558f);
      const float t2 = __builtin_fmaf(kA1, ax, kA0);
      const float den = __builtin_fmaf(q, ax * w, 2.44506634652299f);
      const float n = __builtin_fmaf(t1, q, t2);
      r[i] = (n * __builtin_amdgcn_rcpf(den)) * x;
    }
    else
    {
      static_assert(ACT_T == ACT_TANH, "nam_a1_q_kernel is compiled for Fasttanh and Tanh");
      const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);
      r[i] = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
    }
  }
  return r;
}

// A stage's priority once the launch's first buffer is through it: 0 for every stage (measured: profiles/r05/a1q_variants.txt;
// NAM_AQ_STEADY_PRIO = sixteen values 0 .. 3 for A/B builds)
#ifdef NAM_AQ_STEADY_PRIO
constexpr int kAqSteadyPrio[aq::kNst] = {NAM_AQ_STEADY_PRIO};
#else
constexpr int kAqSteadyPrio[aq::kNst] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
template <int SS>
__device__ __forceinline__ void aq_steady_prio()
{
  if constexpr (kAqSteadyPrio[SS] == 3)
    __builtin_amdgcn_s_setprio((short)3);
  else if constexpr (kAqSteadyPrio[SS] == 2)
    __builtin_amdgcn_s_setprio((short)2);
  else if constexpr (kAqSteadyPrio[SS] == 1)
    __builtin_amdgcn_s_setprio((short)1);
  else
    __builtin_amdgcn_s_setprio((short)0);
}

// The prologue's memory requests in the order the launch's first buffer needs them. A stage's prologue is one memory round trip
// behind the kernel-argument load — 1.7 - 2.5 us when the launch is cold, whatever else is in flight — and 256 workgroups asking
// for all 21 MB of rings at once make that ~4.5 us for everybody. Stage s holds its requests back by s x NAM_AQ_STAG_SLEEP x 64
// cycles (0.64 us per stage: the launch's first buffer reaches stage s ~0.8 s us after stage 0 has started), so stage 0 starts
// on its first sub-block at 2.5 us instead of 4.4 and the other stages' requests land while the buffer is on its way to them
// (profiles/r06/a1q_variants.txt: first output of a 20-buffer launch 26.6 -> 24.0 us; 0 / 3 / 6 / 12 / 24 / 40 measured).
#ifndef NAM_AQ_STAG_SLEEP
#define NAM_AQ_STAG_SLEEP 24
#endif
template <int SS>
__device__ __forceinline__ void aq_stagger()
{
#ifdef NAM_AQ_STAG_FIXED
  if constexpr (SS > 0)
    __builtin_amdgcn_s_sleep(NAM_AQ_STAG_FIXED);
#endif
  if constexpr (SS > 0 && NAM_AQ_STAG_SLEEP > 0)
  {
#pragma unroll 1
    for (int r = 0; r < SS; r++)
      __builtin_amdgcn_s_sleep(NAM_AQ_STAG_SLEEP);
  }
}

namespace aq
{
constexpr int kNotReady = -(1 << 20); // a "consumed" word before its stage is ready (any hand-over wait on it blocks)
constexpr int kNoRow = 1 << 26; // a ring row index no descriptor holds: the access is dropped / returns 0
constexpr int kRowsMax = 1 << 20; // num_records of the ring descriptors (rows)
constexpr int far_jobs_before(int s, int job) // HBM-ring jobs of stage s in front of `job`
{
  int c = 0;
  for (int i = kFirst[s]; i < job; i++)
    c += (has_ring(i) && !res(i)) ? 1 : 0;
  return c;
}
constexpr int max_far_small()
{
  int m = 0;
  for (int s = 0; s < kNst; s++)
    if (is_small(kFirst[s]))
      m = far_jobs_before(s, kFirst[s + 1]) > m ? far_jobs_before(s, kFirst[s + 1]) : m;
  return m;
}
} // namespace aq

// DBG: the profiling instantiation (A1Args::dbg, nam_hip_batch_debug_timeline): workgroup 0's stage s writes row s of the
// buffer — shader-clock stamps at kernel entry, behind the prologue's barrier, at its first and last hand-over, behind its
// write-back; the cycles it spent waiting for input and for its output slot; the buffers it processed.
template <int ACT_T, bool WT, bool PERSIST, bool DBG = false>
__global__ __launch_bounds__(aq::kNst * 64) void nam_a1_q_kernel(const float* __restrict__ blob, const A1Args a)
{
  long long dbg_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if constexpr (DBG)
    dbg_t[0] = clock64();
  using namespace mf;
  using il::kOob;
  using i4 = aq_i4;
  constexpr int NST = aq::kNst;
  extern __shared__ __attribute__((aligned(16))) float lds_aq[];
  char* const lds = reinterpret_cast<char*>(lds_aq);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wall = uni(tid >> 6);
  int S = 0; // this wave's stage
#pragma unroll
  for (int i = 0; i < NST; i++)
    S = wall == i ? aq::kStageOfWave[i] : S;
  // The SIMD's arbiter favours its oldest wave — the EARLIEST stage of the four a SIMD holds. While the launch's first buffer
  // is on its way that is backwards: the later stage has the buffer everything waits for, the earlier one only runs ahead.
  // So the later stages of a SIMD start with the higher priority and every stage drops to 0 once its first buffer is through
  // (first output of a launch 33.6 -> 25.7 us; left on for good the period goes 5.5 -> 6.1 us: profiles/r04/a1q_variants.txt)
  __builtin_amdgcn_s_setprio((short)0);
  if (wall >= 12)
    __builtin_amdgcn_s_setprio((short)3);
  else if (wall >= 8)
    __builtin_amdgcn_s_setprio((short)2);
  else if (wall >= 4)
    __builtin_amdgcn_s_setprio((short)1);
  const int stream = a.stream_map ? a.stream_map[blockIdx.x] : (int)blockIdx.x;
  float* st = a.state + (size_t)stream * a.state_stride;
  const int n_blocks = PERSIST ? (1 << 30) : (a.n_frames + kBlock - 1) / kBlock;

  const int frame = lane; // one lane per frame (stage 0's input sample, the small stages)
  const int g = lane >> 4, n = lane & 15; // the big stages: channel quad g of frames 16 F + n
  const unsigned g16 = (unsigned)g * 16u;
  const unsigned frame16 = (unsigned)frame * 16u;
  const unsigned cls64 = (unsigned)(lane & 3) * 64u; // the lane's record inside an 8 x 8 tile (64 bytes per class)
  const unsigned cls128 = (unsigned)(lane & 3) * 128u; // ... inside a 16 -> 8 tile
  const float* in = a.in ? a.in + (size_t)stream * a.io_stride : nullptr;
  float* out = a.out ? a.out + (size_t)stream * a.io_stride : nullptr;
  const float head_scale = a.head_scale;
  const int io_bytes = PERSIST ? 0x7ffffff0 : a.n_frames * 4;
  const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc((void*)(in ? in : st), 0, in ? io_bytes : 0, 0x00020000);
  const auto rsrc_out = __builtin_amdgcn_make_buffer_rsrc((void*)(out ? out : st), 0, out ? io_bytes : 0, 0x00020000);
  const unsigned long long in_addr = (unsigned long long)(in ? in : st);
  const i4 in_desc = {uni((int)(unsigned)in_addr), uni((int)(unsigned)(in_addr >> 32) & 0xffff), in ? io_bytes : 0, 0x00020000};
  int* wpos_tbl = reinterpret_cast<int*>(st);
  const int wposv = lane < aq::kRings ? wpos_tbl[lane] : 0; // lane r = write position of ring r at launch
  int* const flags = reinterpret_cast<int*>(lds + aq::kFlagB);

  // ---- No workgroup-wide prologue (round 6). Every stage brings in what IT needs — its matrices and constants (registers / its
  // own part of the weight block in LDS), its resident rings — and tells its producer so through its "consumed" word, which
  // starts below zero (kNotReady) and is set to 0 when the stage is ready: the producer's ordinary hand-over wait is the
  // only synchronisation. The launch's first buffer therefore starts when stage 0's 4 KB ring has arrived, not when all
  // 81 KB of the stream's rings have (first hand-over of stage 0 at ~2 us instead of 5.2: profiles/r06/a1q_timeline_*.txt);
  // the later stages' requests are in flight meanwhile and have landed long before the buffer reaches them.
  constexpr int NT = 64;
  auto copy_block = [&](int first_float, int n_floats) { // this wave's part of the weight block -> LDS as it lies (16-byte records)
    const f4* __restrict__ src = reinterpret_cast<const f4*>(blob + a.tiles_off + first_float);
    const int n4 = n_floats / 4;
    for (int i0 = 0; i0 < n4; i0 += 2 * NT)
    {
      const f4 t0 = src[min(i0 + lane, n4 - 1)], t1 = src[min(i0 + NT + lane, n4 - 1)];
      if (i0 + lane < n4)
        lds_st4(lds, (unsigned)aq::kWB + (unsigned)(first_float * 4) + (unsigned)(i0 + lane) * 16u, t0);
      if (i0 + NT + lane < n4)
        lds_st4(lds, (unsigned)aq::kWB + (unsigned)(first_float * 4) + (unsigned)(i0 + NT + lane) * 16u, t1);
    }
  };

  // The stream's rings through two descriptors with the row pitch as the stride (64 / 32 bytes): an access names its row by
  // index and its ring by the scalar offset; kNoRow drops it.
  const unsigned long long st_addr = (unsigned long long)st;
  auto ring_desc = [&](int row_b) {
    return i4{uni((int)(unsigned)st_addr), uni((int)((unsigned)(st_addr >> 32) & 0xffffu) | (row_b << 16)), aq::kRowsMax, 0x00020000};
  };
  const i4 rs16 = ring_desc(aq::kC0 * 4), rs8 = ring_desc(aq::kC1 * 4);
  // row (sb + off) mod R for a wave-uniform sb in [0, R) and a lane offset below 64 <= R
  auto wrap_row = [](int sb, unsigned off, unsigned R) {
    const unsigned v = (unsigned)sb + off;
    return min(v, v - R);
  };
  auto wrap_s = [](int v, int R) { // wave-uniform, v in (-R, R)
    v += v < 0 ? R : 0;
    return v;
  };

  constexpr bool kOutHost = PERSIST && WT; // (kernel_a1_p4.hip: a session whose results go to host memory)
  constexpr int kInAux = PERSIST ? 17 : 0; // session inputs bypass the caches (the caller may rewrite the buffer between commands)
  // Ring appends of a short launch and of a session, and the resident rings' way back into the state when a session's launch
  // leaves, are written through (sc0 sc1): nothing of them is left dirty in the L2 for the end-of-kernel release, which is on the
  // host's critical path when it synchronizes behind a burst (driver-shaped 20-buffer regions 7.85 -> 7.79 us per step in four
  // same-box passes, 500-step regions level: profiles/r05/a1q_variants.txt; NAM_AQ_NO_WT restores round 4's plain stores)
#ifdef NAM_AQ_NO_WT
  constexpr int kAppAux = WT && !PERSIST ? 17 : 0;
  constexpr int kWbAux = 0;
#else
  constexpr int kAppAux = (WT || PERSIST) ? 17 : 0;
  constexpr int kWbAux = PERSIST ? 17 : 0;
#endif
  auto ring_load = [&](unsigned s_) {
    return __hip_atomic_load(a.p_ring + (s_ & (unsigned)a.p_ring_mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  };
  // ---- single-writer words in LDS (kernel_a1_p4.hip): [16 + 2 b] buffers handed over across boundary b, [16 + 2 b + 1]
  // buffers whose LDS reads the consumer has issued; [48], [49] stage 0's first-command decision ----
  const unsigned flag_b = (unsigned)aq::kFlagB;
  auto wait_word = [&](unsigned byte_addr, int want) { // until the word has reached `want` (wave-uniform; wrap-safe)
    // one vector instruction per look (the matrix / vector issue port is what this kernel runs out of): the word goes to the
    // scalar unit, which does the comparison
    int tmp, stmp;
    want = uni(want);
    asm volatile("1:\n\tds_read_b32 %0, %2\n\ts_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %1, %0\n\ts_sub_i32 %1, %1, %3\n\t"
                 "s_cmp_lt_i32 %1, 0\n\ts_cbranch_scc0 2f\n\ts_sleep " NAM_AQ_STR(NAM_AQ_SLEEP) "\n\ts_branch 1b\n2:"
                 : "=&v"(tmp), "=&s"(stmp)
                 : "v"(byte_addr), "s"(want)
                 : "scc");
  };
  auto wait_in = [&](unsigned byte_addr, int want) {
    if constexpr (DBG)
    {
      const long long t0 = clock64();
      wait_word(byte_addr, want);
      dbg_t[5] += clock64() - t0;
    }
    else
      wait_word(byte_addr, want);
  };
  auto wait_out = [&](unsigned byte_addr, int want) {
    if constexpr (DBG)
    {
      const long long t0 = clock64();
      wait_word(byte_addr, want);
      dbg_t[6] += clock64() - t0;
    }
    else
      wait_word(byte_addr, want);
  };
  auto set_word = [&](unsigned byte_addr, int v) {
    asm volatile("" ::: "memory");
    if (lane == 0)
      __hip_atomic_store(reinterpret_cast<int*>(lds + byte_addr), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
  };
  auto prod_b = [&](int b) { return flag_b + (unsigned)(16 + 2 * b) * 4u; };
  auto cons_b = [&](int b) { return flag_b + (unsigned)(16 + 2 * b + 1) * 4u; };

  // a resident ring: stream state <-> LDS planes. 16-channel rings: lane (g, n) moves the quad g of row r0 + n; 8-channel
  // rings: lane (h = lane & 1, r = lane >> 1) the quad h of row r0 + r. Every request of a ring is in flight before the
  // first row is stored (one memory round trip per ring, not one per sixteen rows: a launch of a few buffers pays for it)
  auto ring_to_lds = [&](auto r_tag, int ring_off_f, unsigned lds_b, unsigned plane_b, auto wide_tag) {
    constexpr int R = decltype(r_tag)::value;
    constexpr bool wide = decltype(wide_tag)::value;
    constexpr int per = wide ? 16 : 32, NI = (R + per - 1) / per;
    const int rl = wide ? n : (lane >> 1);
    const unsigned pl = wide ? (unsigned)g : (unsigned)(lane & 1);
    f4 t[NI];
#pragma unroll
    for (int i = 0; i < NI; i++)
    {
      const int row = i * per + rl;
      t[i] = aq_sb_load4(wide ? rs16 : rs8, row < R ? row : aq::kNoRow, (int)(pl * 16u), ring_off_f * 4, 0);
    }
#pragma unroll
    for (int i = 0; i < NI; i++)
    {
      const int row = i * per + rl;
      if (row < R)
        lds_st4(lds, lds_b + pl * plane_b + (unsigned)row * 16u, t[i]);
    }
  };
  auto lds_to_ring = [&](auto r_tag, int ring_off_f, unsigned lds_b, unsigned plane_b, auto wide_tag) {
    constexpr int R = decltype(r_tag)::value;
    constexpr bool wide = decltype(wide_tag)::value;
    constexpr int per = wide ? 16 : 32, NI = (R + per - 1) / per;
    const int rl = wide ? n : (lane >> 1);
    const unsigned pl = wide ? (unsigned)g : (unsigned)(lane & 1);
    f4 t[NI];
#pragma unroll
    for (int i = 0; i < NI; i++)
      t[i] = lds_ld4(lds, lds_b + pl * plane_b + (unsigned)min(i * per + rl, R - 1) * 16u);
#pragma unroll
    for (int i = 0; i < NI; i++)
    {
      const int row = i * per + rl;
      aq_sb_store4(t[i], wide ? rs16 : rs8, row < R ? row : aq::kNoRow, (int)(pl * 16u), ring_off_f * 4, kWbAux);
    }
  };

  unsigned na = 0; // PERSIST, stage 0: commands finished by this stage (its current command carries tag na + 1)
  unsigned done = 0; // PERSIST: commands consumed before this launch (+ finished by the last stage during it)
  unsigned boff0 = 0; // stage 0: byte offset of its first buffer
  if (tid < 64) // words [16 + 2 b + 1], b = 0 .. 14: "consumed" across boundary b — below zero until the consumer is ready
    flags[tid] = (tid >= 17 && tid < 17 + 2 * (NST - 1) && (tid & 1)) ? aq::kNotReady : 0;
  lds_barrier(); // the ONLY workgroup-wide rendezvous of a launch's start: the words are in place
  if constexpr (PERSIST)
  {
    const bool by_value = a.p_seq0 >= 0;
    done = na = by_value ? (unsigned)a.p_seq0 : a.p_cons[blockIdx.x];
    bool ready = true;
    unsigned lo = (unsigned)a.p_cmd0;
    if (!by_value)
    {
      if (wall == 0)
      {
        // started right behind a stream-ordered doorbell on another hardware queue: look for it for a bounded time
        unsigned long long v = session_wait_command<8>(a, ring_load, na + 1u, ring_load(na), (long long)a.p_grace);
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
        session_leaving(a);
        a.p_cons[blockIdx.x] = done;
        __hip_atomic_store(a.p_done + blockIdx.x, done | 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      return;
    }
    boff0 = lo * 4u;
  }

  int nvalid = kBlock; // frames of the buffer this wave's stage is working on
  bool more = false; // this stage has another buffer behind the current one
  unsigned boff = 0; // byte offset of the current buffer in the stream's row
  // write positions go back into the state: lane r = ring r, for the rings [r0, r0 + nr) whose positions are wpv[0 .. nr)
  auto store_positions = [&](int r0, int nr, const int* wpv) {
    int v = 0;
#pragma unroll
    for (int u = 0; u < 5; u++)
      v = (u < nr && lane == r0 + u) ? wpv[u < nr ? u : 0] : v;
    if (lane >= r0 && lane < r0 + nr)
      wpos_tbl[lane] = v;
  };

  // ==============================================================================================
  // BIG stages (array 0): jobs J0 .. J0 + NJS - 1 are layers of 16 channels. A buffer flows through them as FOUR SUB-BLOCKS
  // of 16 frames — lane (g, n) holds channels 4 g .. 4 g + 3 of frame 16 i + n of sub-block i —, each handed to the next stage
  // as soon as it is done: the first buffer of a launch (and every buffer of a lone caller) is in four big stages at once
  // instead of one, the stage bodies are a quarter of the code, and the words "produced" / "consumed" count sub-blocks.
  // ==============================================================================================
  auto run_big = [&](auto s_tag) {
    constexpr int SS = decltype(s_tag)::value;
    constexpr int J0 = aq::kFirst[SS], NJS = aq::kFirst[SS + 1] - J0, JN = aq::kFirst[SS + 1];
    constexpr bool FIRST = SS == 0;
    constexpr int QIN = SS - 1, QOUT = SS;
    constexpr int NFAR = aq::far_jobs_before(SS, JN);
    // how many sub-blocks the slot (and a non-resident input area) holds: two into a big stage, the whole buffer into the
    // transition, which takes it at once
    constexpr int DIN = aq::depth_in(J0), DOUT = JN == aq::kJobT ? 4 : aq::depth_in(JN);
    static_assert(aq::is_big(JN) || JN == aq::kJobT, "aq: array 0's last stage hands over to the transition");
    // ---- this stage's weights: registers for the whole launch. Tile q of job j (plan_a1.cpp: build_a1_ws, FULL layout):
    // lane (g, i) holds W[out = i][in = 4 g + s], s = 0 .. 3 — the A operand of k-step s ----
    aq_stagger<SS>();
    f4 W[NJS][4];
    {
      const f4* __restrict__ tsrc = reinterpret_cast<const f4*>(blob + a.consts_off); // (the p2 tiles: A1Args::consts_off, see launch_a1_q)
#pragma unroll
      for (int u = 0; u < NJS; u++)
#pragma unroll
        for (int q = 0; q < 4; q++)
          W[u][q] = tsrc[(J0 + u) * 256 + q * 64 + lane];
    }
    int wp[NJS];
#pragma unroll
    for (int u = 0; u < NJS; u++)
      wp[u] = __builtin_amdgcn_readlane(wposv, aq::ring_id(J0 + u));
    int wpo = aq::res(JN) ? __builtin_amdgcn_readlane(wposv, aq::ring_id(aq::res(JN) ? JN : 0)) : 0; // the next stage's first ring
    // lane part of an LDS row address per ring of this stage (+ the next stage's input area)
    unsigned gb[NJS];
#pragma unroll
    for (int u = 0; u < NJS; u++)
      gb[u] = (unsigned)aq::in_b(J0 + u) + (unsigned)g * (unsigned)aq::plane_b(J0 + u);
    const unsigned gbn = (unsigned)aq::in_b(JN) + (unsigned)g * (unsigned)aq::plane_b(JN);
    // resident rings of this stage: state -> LDS
    il::for_each_index(
      [&](auto u_tag) {
        constexpr int TJ = J0 + decltype(u_tag)::value;
        if constexpr (aq::res(TJ))
          ring_to_lds(std::integral_constant<int, aq::ring_len(TJ)>{}, aq::ring_off(TJ), (unsigned)aq::in_b(TJ), (unsigned)aq::plane_b(TJ), std::true_type{});
      },
      std::make_integer_sequence<int, NJS>{});
    // HBM rings: the far taps' rows, requested ONE SUB-BLOCK ahead (every one of them is at least two buffers old)
    f4 far[NFAR > 0 ? NFAR : 1][2];
    auto fetch_far = [&](auto tj_tag, int sb) { // rows (sb + n - L) mod R of ring TJ, L = 2 d, d; sb = position of the sub-block's first frame
      constexpr int TJ = decltype(tj_tag)::value;
      constexpr int R = aq::ring_len(TJ), D = aq::dil(TJ), FI = aq::far_jobs_before(SS, TJ);
#pragma unroll
      for (int j = 0; j < 2; j++)
        far[FI][j] = aq_sb_load4(rs16, (int)wrap_row(wrap_s(sb - (2 - j) * D, R), (unsigned)n, R), (int)g16, aq::ring_off(TJ) * 4, 0);
    };
    il::for_each_index(
      [&](auto u_tag) {
        constexpr int TJ = J0 + decltype(u_tag)::value;
        if constexpr (!aq::res(TJ))
          fetch_far(std::integral_constant<int, TJ>{}, wp[TJ - J0]);
      },
      std::make_integer_sequence<int, NJS>{});
    float inp = 0.0f; // stage 0: the next buffer's input sample of frame `lane`
    if constexpr (FIRST)
      inp = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_in, frame * 4, uni((int)boff0), kInAux));
    // the layers' constants (bias, mixin, 1x1 bias by channel quad): registers for the whole launch, like the matrices
    // (same-box A/B, profiles/r04/a1q_variants.txt: 5.63 -> 5.37 us per buffer inside a 200-buffer launch); straight from the
    // weight block in memory (no other wave's copy to wait for)
    f4 cst[NJS][3];
    f4 rech = {0.f, 0.f, 0.f, 0.f}; // stage 0: array 0's rechannel column, this lane's channel quad
    {
      const f4* __restrict__ csrc = reinterpret_cast<const f4*>(blob + a.tiles_off + aq::kBigConsts);
#pragma unroll
      for (int u = 0; u < NJS; u++)
#pragma unroll
        for (int q = 0; q < 3; q++)
          cst[u][q] = csrc[(J0 + u) * 16 + q * 4 + g];
      if constexpr (FIRST)
        rech = csrc[12 + g];
    }
    // this stage is ready: its rings are in LDS (every store of the prologue has been executed), its producer may hand over
    __builtin_amdgcn_s_waitcnt(0);
    if constexpr (!FIRST)
      set_word(cons_b(QIN), 0);
    if constexpr (DBG)
      dbg_t[1] = clock64();

    f4 xs, hd;
    float cnd = 0.0f;
    unsigned long long spec_cmd = 0; // PERSIST, stage 0: the early look at the next command ...
    float inp_spec = 0.0f; // ... and the input sample requested on a hit
    int k = 0; // buffers finished by this stage
    int m = 0; // sub-blocks finished by this stage (4 k + i, modulo 2^32: the words are compared through differences)
    int i = 0; // sub-block of the buffer

    // One layer on one sub-block: z = act(conv(x) + mixin(cond)); head += z; x += layer1x1(z)   (model.cpp:183-393)
    auto job = [&](auto j_tag) {
      constexpr int JI = decltype(j_tag)::value;
      constexpr int U = JI - J0;
      constexpr int R = aq::ring_len(JI), D = aq::dil(JI);
      constexpr bool RES = aq::res(JI);
      constexpr bool TAKES = U == 0 && SS > 0; // the input rows come from the previous stage (LDS)
      constexpr int FI = aq::far_jobs_before(SS, JI);
      __builtin_amdgcn_sched_barrier(0);
      const int wpj = wp[U];
      int sb = wpj + 16 * i; // ring position of the sub-block's first frame (scalar unit)
      sb -= sb >= R ? R : 0;
      // (a) the sub-block's input rows
      const unsigned row_cur = (RES || !TAKES) ? wrap_row(sb, (unsigned)n, R) : (unsigned)(16 * (m & (DIN - 1)) + n);
      if constexpr (TAKES)
        xs = lds_ld4(lds, gb[U] + row_cur * 16u);
      else if constexpr (RES)
      {
        lds_st4(lds, gb[U] + row_cur * 16u, xs);
        asm volatile("" ::: "memory"); // the taps read OTHER lanes' rows: not above this store
      }
      if constexpr (!RES)
        aq_sb_store4(xs, rs16, 16 * i + n < nvalid ? (int)wrap_row(sb, (unsigned)n, R) : aq::kNoRow, (int)g16, aq::ring_off(JI) * 4, kAppAux);
      // (b) the taps' rows: LDS ring, or the registers requested a sub-block ago
      f4 bt[2];
#pragma unroll
      for (int j = 0; j < 2; j++)
      {
        if constexpr (RES)
          bt[j] = lds_ld4(lds, gb[U] + wrap_row(wrap_s(sb - (2 - j) * D, R), (unsigned)n, R) * 16u);
        else
          bt[j] = far[FI][j];
      }
      if constexpr (TAKES)
        set_word(cons_b(QIN), m + 1); // every LDS read of what the previous stage may overwrite next is issued
      // persistent session, stage 0: look at the next ring slot in the buffer's first sub-block and, when the command is
      // already there, request the next buffer's input sample from it in the third (kernel_kq.hip)
      if constexpr (PERSIST && JI == 0)
      {
        if (i == 0)
        {
          const unsigned long long v = ring_load(na + 1);
          spec_cmd = ((unsigned long long)(unsigned)uni((int)(unsigned)(v >> 32)) << 32) | (unsigned long long)(unsigned)uni((int)(unsigned)v);
        }
        if (i == 2)
        {
          const bool hit = (unsigned)(spec_cmd >> 32) == na + 2;
          const int soff = uni(hit ? (int)((unsigned)spec_cmd * 4u) : 0);
          inp_spec = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_in, hit ? frame * 4 : (int)kOob, soff, kInAux));
        }
      }
      // (c) conv + mixin: one chain seeded with bias + mixin * input, taps oldest first (two chains and an add — a lone
      // wave's dependent MFMA waits 40 cycles, issue is 32 — measured slower: profiles/r04/a1q_variants.txt)
      const f4 bv = cst[U][0], mv = cst[U][1];
      f4 acc = __builtin_elementwise_fma(mv, f4{cnd, cnd, cnd, cnd}, bv);
#pragma unroll
      for (int s_ = 0; s_ < 4; s_++)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(W[U][0][s_], bt[0][s_], acc, 0, 0, 0);
#pragma unroll
      for (int s_ = 0; s_ < 4; s_++)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(W[U][1][s_], bt[1][s_], acc, 0, 0, 0);
      // the far taps' rows of the NEXT sub-block, into the registers just consumed
      if constexpr (!RES)
      {
        int wpn_ = wpj + nvalid;
        wpn_ -= wpn_ >= R ? R : 0;
        int nsb = sb + 16;
        nsb -= nsb >= R ? R : 0;
        fetch_far(j_tag, i == 3 ? wpn_ : nsb);
      }
#pragma unroll
      for (int s_ = 0; s_ < 4; s_++)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(W[U][2][s_], xs[s_], acc, 0, 0, 0);
      // (d) activation, head accumulator, layer 1x1 + residual
      const f4 b1v = cst[U][2];
      const f4 z = aq_act4<ACT_T>(acc);
      hd += z;
      xs += b1v;
#pragma unroll
      for (int s_ = 0; s_ < 4; s_++)
        xs = __builtin_amdgcn_mfma_f32_16x16x4f32(W[U][3][s_], z[s_], xs, 0, 0, 0);
    };

    boff = boff0;
    bool have = !FIRST || n_blocks > 0;
#pragma unroll 1
    for (;; k++)
    {
      bool exit_tok = false;
      float cond = 0.0f;
      constexpr unsigned islot = (unsigned)aq::slot_b(QIN < 0 ? 0 : QIN), oslot = (unsigned)aq::slot_b(QOUT);
      if constexpr (FIRST)
      {
        exit_tok = !have;
        nvalid = PERSIST ? kBlock : min(kBlock, a.n_frames - k * kBlock);
        more = PERSIST || k + 1 < n_blocks;
      }
      else
      {
        // the token of buffer k travels with its first sub-block
        wait_in(prod_b(QIN), m + 1);
        asm volatile("" ::: "memory");
        const i4 tok = *reinterpret_cast<const i4*>(lds + islot + (unsigned)(DIN * 1024 + DIN * 64));
        boff = (unsigned)uni(tok[0]);
        nvalid = uni(tok[1]);
        exit_tok = uni(tok[2]) != 0;
        more = PERSIST || uni(tok[3]) != 0;
      }
      if (exit_tok)
      {
        wait_out(cons_b(QOUT), m - (DOUT - 1));
        asm volatile("" ::: "memory");
        if (lane == 0)
          *reinterpret_cast<i4*>(lds + oslot + (unsigned)(DOUT * 1024 + DOUT * 64)) = i4{(int)boff, nvalid, 1, 0};
        set_word(prod_b(QOUT), m + 1);
        break;
      }
      if constexpr (FIRST)
      {
        cond = inp; // this buffer's input sample (requested a buffer ago)
        if constexpr (!PERSIST) // next block's (offset beyond the launch's frames -> 0)
          inp = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_in, frame * 4, uni((k + 1) * (kBlock * 4)), 0));
      }
#pragma unroll 1
      for (i = 0; i < 4; i++)
      {
        if constexpr (FIRST)
        {
          // array 0's rechannel (1 -> 16, model.cpp:488-490): x = column * input
          cnd = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((16 * i + n) * 4, __builtin_bit_cast(int, cond)));
          xs = rech * cnd;
          hd = f4{0.f, 0.f, 0.f, 0.f};
        }
        else
        {
          if (i > 0)
          {
            wait_in(prod_b(QIN), m + 1);
            asm volatile("" ::: "memory");
          }
          const unsigned qr = (unsigned)(16 * (m & (DIN - 1)) + n); // the sub-block's rows in the slot
          hd = lds_ld4(lds, islot + g16 * (unsigned)(DIN * 16) + qr * 16u);
          cnd = *reinterpret_cast<const float*>(lds + islot + (unsigned)(DIN * 1024) + qr * 4u);
        }
        il::for_each_index([&](auto u_tag) { job(std::integral_constant<int, J0 + decltype(u_tag)::value>{}); },
                           std::make_integer_sequence<int, NJS>{});
        // hand-over of the sub-block: its rows into the next stage's first ring (or input area), the rest into the slot; the
        // consumer has issued every LDS read of the sub-block DOUT back (whose slot rows — and, four back, ring rows — these
        // overwrite)
        wait_out(cons_b(QOUT), m - (DOUT - 1));
        asm volatile("" ::: "memory");
        {
          int so = wpo + 16 * i;
          so -= (aq::res(JN) && so >= aq::ring_len(JN)) ? aq::ring_len(JN) : 0;
          const unsigned qr = (unsigned)(16 * (m & (DOUT - 1)) + n);
          const unsigned row = aq::res(JN) ? wrap_row(so, (unsigned)n, aq::ring_len(JN)) : qr;
          lds_st4(lds, gbn + row * 16u, xs);
          lds_st4(lds, oslot + g16 * (unsigned)(DOUT * 16) + qr * 16u, hd);
          if (g == 0)
            *reinterpret_cast<float*>(lds + oslot + (unsigned)(DOUT * 1024) + qr * 4u) = cnd;
          if (i == 0 && lane == 0)
            *reinterpret_cast<i4*>(lds + oslot + (unsigned)(DOUT * 1024 + DOUT * 64)) = i4{(int)boff, nvalid, 0, more ? 1 : 0};
        }
        set_word(prod_b(QOUT), m + 1);
        m = (int)((unsigned)m + 1u);
        if constexpr (DBG)
        {
          dbg_t[3] = clock64();
          dbg_t[2] = dbg_t[2] ? dbg_t[2] : dbg_t[3];
          dbg_t[7] = m;
        }
      }
      if (k == 0)
        aq_steady_prio<SS>(); // (the launch's first buffer is through this stage)
      // the rings move on by the buffer's frames (scalar unit)
#pragma unroll
      for (int u = 0; u < NJS; u++)
      {
        int np = wp[u] + nvalid;
        np -= np >= aq::ring_len(J0 + u) ? aq::ring_len(J0 + u) : 0;
        wp[u] = np;
      }
      if constexpr (aq::res(JN))
      {
        wpo += nvalid;
        wpo -= wpo >= aq::ring_len(JN) ? aq::ring_len(JN) : 0;
      }
      if constexpr (FIRST)
      {
        // the next buffer of this stage (kernel_kq.hip)
        if constexpr (PERSIST)
        {
          const unsigned tag = na + 2u; // the command behind the one just finished
          unsigned long long v = spec_cmd;
          if ((unsigned)(v >> 32) != tag)
          {
            v = session_wait_command<16>(a, ring_load, tag, ring_load(tag - 1u), (long long)(a.p_linger > 0 ? a.p_linger : 100)); // (1 us unless the launch lingers)
          }
          // ONE view of the ring slot for the whole wave (lane 0's)
          const unsigned v_tag = (unsigned)uni((int)(unsigned)(v >> 32)), v_off = (unsigned)uni((int)(unsigned)v);
          have = v_tag == tag;
          if (!have && lane == 0)
            session_leaving(a); // (the others stop waiting for workgroups behind them: il_common.h)
          const unsigned next_off = v_off * 4u;
          na++;
          if (have)
          {
            boff = next_off;
            const bool mine = (unsigned)(spec_cmd >> 32) == na + 1u && (unsigned)spec_cmd * 4u == next_off;
            inp = inp_spec;
            if (!mine)
            {
              const int voff = frame * 4, soff = uni((int)next_off);
              const i4 rsd = in_desc;
              asm volatile("buffer_load_dword %0, %1, %2, %3 offen sc0 sc1\n\ts_waitcnt vmcnt(0)"
                           : "=v"(inp)
                           : "v"(voff), "s"(rsd), "s"(soff)
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
    // the launch leaves: resident rings and write positions go back into the state
    asm volatile("" ::: "memory");
    il::for_each_index(
      [&](auto u_tag) {
        constexpr int TJ = J0 + decltype(u_tag)::value;
        if constexpr (aq::res(TJ))
          lds_to_ring(std::integral_constant<int, aq::ring_len(TJ)>{}, aq::ring_off(TJ), (unsigned)aq::in_b(TJ), (unsigned)aq::plane_b(TJ), std::true_type{});
      },
      std::make_integer_sequence<int, NJS>{});
    store_positions(aq::ring_id(J0), NJS, wp);
  };

  // ==============================================================================================
  // SMALL stages (array 1, 8 channels): one lane per frame, 64 frames at a time (kernel_kq.hip's job on rings). The first of
  // them starts with the TRANSITION (job 10): array 1's rechannel and array 0's head rechannel, 16 -> 8 each (model.cpp:476-490:
  // array 1's head accumulator starts as array 0's head output; its layers run on rechannel(x)); the last one ends with the
  // head (job 21).
  // ==============================================================================================
  auto run_small = [&](auto s_tag) {
    constexpr int SS = decltype(s_tag)::value;
    constexpr int J0 = aq::kFirst[SS], JE = aq::kFirst[SS + 1];
    constexpr bool LAST = SS == NST - 1, HAS_T = J0 == aq::kJobT;
    constexpr int JM0 = HAS_T ? J0 + 1 : J0; // the stage's first layer
    constexpr int NL = (LAST ? JE - 1 : JE) - JM0; // layers (the last stage's final job is the head)
    constexpr int JN = LAST ? JM0 : JE; // the next stage's first job (a small layer)
    constexpr int QIN = SS - 1, QOUT = SS;
    constexpr int NFAR = aq::far_jobs_before(SS, JM0 + NL);
    static_assert(NL >= 1 && (LAST || aq::is_small(JN)), "aq: small stages hold layers");
    aq_stagger<SS>();
    int wp[NL];
#pragma unroll
    for (int u = 0; u < NL; u++)
      wp[u] = __builtin_amdgcn_readlane(wposv, aq::ring_id(JM0 + u));
    int wpo = (!LAST && aq::res(JN)) ? __builtin_amdgcn_readlane(wposv, aq::ring_id(JN)) : 0;
    il::for_each_index(
      [&](auto u_tag) {
        constexpr int TJ = JM0 + decltype(u_tag)::value;
        if constexpr (aq::res(TJ))
          ring_to_lds(std::integral_constant<int, aq::ring_len(TJ)>{}, aq::ring_off(TJ), (unsigned)aq::in_b(TJ), (unsigned)aq::plane_b(TJ), std::false_type{});
      },
      std::make_integer_sequence<int, NL>{});
    struct Row
    {
      f4 q0, q1;
    };
    Row far[NFAR > 0 ? NFAR : 1][2];
    auto fetch_far = [&](auto tj_tag, int wpj) {
      constexpr int TJ = decltype(tj_tag)::value;
      constexpr int R = aq::ring_len(TJ), D = aq::dil(TJ), FI = aq::far_jobs_before(SS, TJ);
#pragma unroll
      for (int j = 0; j < 2; j++)
      {
        const int idx = (int)wrap_row(wrap_s(wpj - (2 - j) * D, R), (unsigned)frame, R);
        far[FI][j].q0 = aq_sb_load4(rs8, idx, 0, aq::ring_off(TJ) * 4, 0);
        far[FI][j].q1 = aq_sb_load4(rs8, idx, 16, aq::ring_off(TJ) * 4, 0);
      }
    };
    il::for_each_index(
      [&](auto u_tag) {
        constexpr int TJ = JM0 + decltype(u_tag)::value;
        if constexpr (!aq::res(TJ))
          fetch_far(std::integral_constant<int, TJ>{}, wp[TJ - JM0]);
      },
      std::make_integer_sequence<int, NL>{});
    // this stage's own part of the weight block -> LDS: its layers' tiles and constants (+ the transition's / the head's)
    copy_block(aq::kMTiles + (JM0 - aq::kJobM0) * 4 * aq::kTileM, NL * 4 * aq::kTileM);
    copy_block(aq::kMConsts + (JM0 - aq::kJobM0) * 24, NL * 24);
    if constexpr (HAS_T)
      copy_block(aq::kWrOff, 2 * aq::kTileT);
    if constexpr (LAST)
      copy_block(aq::kHeadTile, aq::kTileM);
    if constexpr (HAS_T || LAST)
      copy_block(aq::kTConsts, 16);
    __builtin_amdgcn_s_waitcnt(0);
    set_word(cons_b(QIN), 0); // ready: the producer may hand over
    if constexpr (DBG)
      dbg_t[1] = clock64();

    f4 x0, x1, head0, head1;
    float cond = 0.0f;
    int k = 0;
    auto job = [&](auto j_tag) {
      constexpr int JI = decltype(j_tag)::value;
      constexpr int U = JI - JM0, LI = JI - aq::kJobM0;
      constexpr int R = aq::ring_len(JI), D = aq::dil(JI);
      constexpr bool RES = aq::res(JI), TAKES = U == 0 && !HAS_T; // the input rows come from the previous stage (LDS)
      constexpr int FI = aq::far_jobs_before(SS, JI);
      constexpr unsigned rb0 = (unsigned)aq::in_b(JI), rb1 = rb0 + (unsigned)aq::plane_b(JI);
      __builtin_amdgcn_sched_barrier(0);
      const int wpj = wp[U];
      const unsigned row0 = wrap_row(wpj, (unsigned)frame, R);
      if constexpr (TAKES)
      {
        const unsigned r_in = RES ? row0 * 16u : frame16; // (an HBM-ring layer takes its rows through a 64-row area)
        x0 = lds_ld4(lds, rb0 + r_in);
        x1 = lds_ld4(lds, rb1 + r_in);
      }
      else if constexpr (RES)
      {
        lds_st4(lds, rb0 + row0 * 16u, x0);
        lds_st4(lds, rb1 + row0 * 16u, x1);
        asm volatile("" ::: "memory");
      }
      if constexpr (!RES)
      {
        const int widx = frame < nvalid ? (int)row0 : aq::kNoRow;
        aq_sb_store4(x0, rs8, widx, 0, aq::ring_off(JI) * 4, kAppAux);
        aq_sb_store4(x1, rs8, widx, 16, aq::ring_off(JI) * 4, kAppAux);
      }
      int wpn = wpj + nvalid;
      wpn -= wpn >= R ? R : 0;
      f4 b0[2], b1[2];
#pragma unroll
      for (int j = 0; j < 2; j++)
      {
        if constexpr (RES)
        {
          const unsigned row = wrap_row(wrap_s(wpj - (2 - j) * D, R), (unsigned)frame, R) * 16u;
          b0[j] = lds_ld4(lds, rb0 + row);
          b1[j] = lds_ld4(lds, rb1 + row);
        }
        else
          b0[j] = far[FI][j].q0, b1[j] = far[FI][j].q1;
      }
      if constexpr (TAKES)
        set_word(cons_b(QIN), k + 1);
      constexpr unsigned cb = (unsigned)(aq::kWB + (aq::kMConsts + LI * 24) * 4);
      f4 acc0 = lds_ld4(lds, cb), acc1 = lds_ld4(lds, cb + 16u);
      {
        const f4 m0 = lds_ld4(lds, cb + 32u), m1 = lds_ld4(lds, cb + 48u);
        const f4 c4 = {cond, cond, cond, cond};
        acc0 = __builtin_elementwise_fma(m0, c4, acc0);
        acc1 = __builtin_elementwise_fma(m1, c4, acc1);
      }
      // a tap = 16 matrix instructions on a 256-byte tile; the NEXT tap's tile is requested before this tap's instructions are
      // issued (a lone wave — the first buffer of a launch, a one-stream caller — otherwise sits out an LDS round trip per tap)
      struct Tile
      {
        f4 a, b, c, d;
      };
      auto tile_ld = [&](unsigned tb) { return Tile{lds_ld4(lds, tb + cls64), lds_ld4(lds, tb + 16u + cls64), lds_ld4(lds, tb + 32u + cls64), lds_ld4(lds, tb + 48u + cls64)}; };
      auto tap = [&](const Tile& w, const f4& v0, const f4& v1, f4& o0, f4& o1) {
#pragma unroll
        for (int c = 0; c < 4; c++)
        {
          o0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w.a[c], v0[c], o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w.c[c], v0[c], o1, 0, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < 4; c++)
        {
          o0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w.b[c], v1[c], o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w.d[c], v1[c], o1, 0, 0, 0);
        }
      };
      constexpr unsigned t0 = (unsigned)(aq::kWB + (aq::kMTiles + LI * 4 * aq::kTileM) * 4);
      const Tile w0 = tile_ld(t0), w1 = tile_ld(t0 + 256u);
      tap(w0, b0[0], b1[0], acc0, acc1);
      const Tile w2 = tile_ld(t0 + 512u);
      tap(w1, b0[1], b1[1], acc0, acc1);
      if constexpr (!RES)
        fetch_far(j_tag, wpn);
      const Tile w3 = tile_ld(t0 + 768u);
      tap(w2, x0, x1, acc0, acc1);
      const f4 b1v0 = lds_ld4(lds, cb + 64u), b1v1 = lds_ld4(lds, cb + 80u);
      const f4 z0 = aq_act4<ACT_T>(acc0), z1 = aq_act4<ACT_T>(acc1);
      head0 += z0;
      head1 += z1;
      f4 y0 = x0 + b1v0, y1 = x1 + b1v1;
      tap(w3, z0, z1, y0, y1);
      x0 = y0;
      x1 = y1;
      wp[U] = wpn;
    };

#pragma unroll 1
    for (;; k++)
    {
      constexpr unsigned slot = (unsigned)aq::slot_b(QIN), oslot = (unsigned)aq::slot_b(LAST ? QIN : QOUT);
      i4 tok;
      bool exit_tok;
      if constexpr (HAS_T)
      {
        // array 0's last stage hands over sub-blocks: the token comes with the first, the buffer is whole with the fourth
        constexpr unsigned xin = (unsigned)aq::in_b(aq::kJobT);
        const int m0 = (int)(4u * (unsigned)k);
        wait_in(prod_b(QIN), m0 + 1);
        asm volatile("" ::: "memory");
        tok = *reinterpret_cast<const i4*>(lds + slot + 4352u);
        exit_tok = uni(tok[2]) != 0;
        if (!exit_tok)
          wait_in(prod_b(QIN), m0 + 4);
        asm volatile("" ::: "memory");
        f4 x16[4], h16[4];
#pragma unroll
        for (int p = 0; p < 4; p++)
        {
          x16[p] = lds_ld4(lds, xin + (unsigned)p * 1024u + frame16);
          h16[p] = lds_ld4(lds, slot + (unsigned)p * 1024u + frame16);
        }
        cond = *reinterpret_cast<const float*>(lds + slot + 4096u + (unsigned)lane * 4u);
        set_word(cons_b(QIN), m0 + 4);
        // x8 = Wr x16; h8 = bias + Wh h16: per output half one chain, input channels in order
        constexpr unsigned wr = (unsigned)(aq::kWB + aq::kWrOff * 4), wh = (unsigned)(aq::kWB + aq::kWhOff * 4), tc = (unsigned)(aq::kWB + aq::kTConsts * 4);
        x0 = x1 = f4{0.f, 0.f, 0.f, 0.f};
        head0 = lds_ld4(lds, tc);
        head1 = lds_ld4(lds, tc + 16u);
#pragma unroll
        for (int p = 0; p < 4; p++)
        {
          const f4 w0 = lds_ld4(lds, wr + cls128 + (unsigned)p * 16u), w1 = lds_ld4(lds, wr + cls128 + 64u + (unsigned)p * 16u);
#pragma unroll
          for (int c = 0; c < 4; c++)
          {
            x0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w0[c], x16[p][c], x0, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w1[c], x16[p][c], x1, 0, 0, 0);
          }
        }
#pragma unroll
        for (int p = 0; p < 4; p++)
        {
          const f4 w0 = lds_ld4(lds, wh + cls128 + (unsigned)p * 16u), w1 = lds_ld4(lds, wh + cls128 + 64u + (unsigned)p * 16u);
#pragma unroll
          for (int c = 0; c < 4; c++)
          {
            head0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w0[c], h16[p][c], head0, 0, 0, 0);
            head1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w1[c], h16[p][c], head1, 0, 0, 0);
          }
        }
      }
      else
      {
        wait_in(prod_b(QIN), k + 1);
        asm volatile("" ::: "memory");
        tok = *reinterpret_cast<const i4*>(lds + slot + 2304u);
        head0 = lds_ld4(lds, slot + frame16);
        head1 = lds_ld4(lds, slot + 1024u + frame16);
        cond = *reinterpret_cast<const float*>(lds + slot + 2048u + (unsigned)lane * 4u);
        exit_tok = uni(tok[2]) != 0;
      }
      boff = (unsigned)uni(tok[0]);
      nvalid = uni(tok[1]);
      if (exit_tok)
      {
        if constexpr (!HAS_T)
          set_word(cons_b(QIN), k + 1);
        if constexpr (!LAST)
        {
          wait_out(cons_b(QOUT), k);
          asm volatile("" ::: "memory");
          if (lane == 0)
            *reinterpret_cast<i4*>(lds + oslot + 2304u) = tok;
          set_word(prod_b(QOUT), k + 1);
        }
        break;
      }
      il::for_each_index([&](auto u_tag) { job(std::integral_constant<int, JM0 + decltype(u_tag)::value>{}); },
                         std::make_integer_sequence<int, NL>{});
      if constexpr (DBG)
      {
        dbg_t[3] = clock64();
        dbg_t[2] = dbg_t[2] ? dbg_t[2] : dbg_t[3];
        dbg_t[7] = k + 1;
      }
      if (k == 0)
        aq_steady_prio<SS>(); // (the launch's first buffer is through this stage's layers)
      if constexpr (!LAST)
      {
        constexpr int RN = aq::ring_len(JN);
        wait_out(cons_b(QOUT), k);
        asm volatile("" ::: "memory");
        const unsigned row = aq::res(JN) ? wrap_row(wpo, (unsigned)frame, RN) * 16u : frame16;
        lds_st4(lds, (unsigned)aq::in_b(JN) + row, x0);
        lds_st4(lds, (unsigned)(aq::in_b(JN) + aq::plane_b(JN)) + row, x1);
        lds_st4(lds, oslot + frame16, head0);
        lds_st4(lds, oslot + 1024u + frame16, head1);
        *reinterpret_cast<float*>(lds + oslot + 2048u + (unsigned)lane * 4u) = cond;
        if (lane == 0)
          *reinterpret_cast<i4*>(lds + oslot + 2304u) = tok;
        if constexpr (aq::res(JN))
        {
          wpo += nvalid;
          wpo -= wpo >= RN ? RN : 0;
        }
        set_word(prod_b(QOUT), k + 1);
      }
      else
      {
        // array 1's head rechannel (8 -> 1, bias), head_scale (model.cpp:547-549, 886-910)
        constexpr unsigned tb = (unsigned)(aq::kWB + aq::kHeadTile * 4);
        const f4 wa = lds_ld4(lds, tb + cls64), wb = lds_ld4(lds, tb + 16u + cls64);
        f4 acc = lds_ld4(lds, (unsigned)(aq::kWB + aq::kTConsts * 4) + 32u);
#pragma unroll
        for (int c = 0; c < 4; c++)
          acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wa[c], head0[c], acc, 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 4; c++)
          acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wb[c], head1[c], acc, 0, 0, 0);
        const float yout = head_scale * acc[0];
        if (kOutHost && a.p_out_host == 2) // ticketed host buffers (nam_hip_batch_submit_f32): written through, see below
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yout), rsrc_out, frame < nvalid ? frame * 4 : (int)kOob, uni((int)boff), 17);
        else
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yout), rsrc_out, frame < nvalid ? frame * 4 : (int)kOob, uni((int)boff),
                                                PERSIST && !kOutHost ? 17 : 0);
        if constexpr (PERSIST)
        {
          done++;
          if (kOutHost && a.p_out_host == 2)
          {
            // every command's results are a completion of their own: the system-scope stores above have been acknowledged
            // when vmcnt reaches 0; then the workgroup counts itself in (kernels.h: p_cmd_count / p_cmd_done)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned before = 0u;
            const unsigned cslot = (done - 1u) & (unsigned)a.p_ring_mask;
            if (lane == 0)
              before = __hip_atomic_fetch_add(a.p_cmd_count + cslot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)uni((int)before) == gridDim.x - 1u && lane == 0)
            {
              // the last workgroup through command `done - 1`: every other one had its results acknowledged before it counted
              __hip_atomic_store(a.p_cmd_count + cslot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(a.p_cmd_done + cslot, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
              __hip_atomic_fetch_max(a.p_cmd_count + a.p_ring_mask + 1, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (il_common.h: session_wait_command)
            }
            if (lane == 0 && (done & 15u) == 0u)
              __hip_atomic_store(a.p_prog + blockIdx.x, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
          else if (lane == 0 && (done & 15u) == 0u) // progress for the host's ring bookkeeping (not a completion signal)
            __hip_atomic_store(a.p_prog + blockIdx.x, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
    asm volatile("" ::: "memory");
    il::for_each_index(
      [&](auto u_tag) {
        constexpr int TJ = JM0 + decltype(u_tag)::value;
        if constexpr (aq::res(TJ))
          lds_to_ring(std::integral_constant<int, aq::ring_len(TJ)>{}, aq::ring_off(TJ), (unsigned)aq::in_b(TJ), (unsigned)aq::plane_b(TJ), std::false_type{});
      },
      std::make_integer_sequence<int, NL>{});
    store_positions(aq::ring_id(JM0), NL, wp);
  };

  il::for_each_index(
    [&](auto s_tag) {
      constexpr int SS = decltype(s_tag)::value;
      if (S == SS)
      {
        if constexpr (aq::is_big(aq::kFirst[SS]))
          run_big(s_tag);
        else
          run_small(s_tag);
      }
    },
    std::make_integer_sequence<int, NST>{});

  if constexpr (DBG)
  {
    dbg_t[4] = clock64();
    if (a.dbg && blockIdx.x == 0 && lane < 8)
    {
      long long v = 0;
#pragma unroll
      for (int i = 0; i < 8; i++)
        v = lane == i ? dbg_t[i] : v;
      a.dbg[S * 8 + lane] = v;
    }
  }
  if constexpr (PERSIST)
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    if constexpr (kOutHost)
    {
      if (S == NST - 1) // one wave per workgroup asks for the write-back (kernel_a1_p4.hip)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    }
    if (S == NST - 1 && lane == 0)
    {
      a.p_cons[blockIdx.x] = done;
      __hip_atomic_store(a.p_done + blockIdx.x, done | 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

#ifdef NAM_AQ_PROBE // developer builds (ISA inspection): one instantiation only
template __global__ void nam_a1_q_kernel<ACT_FASTTANH, false, true>(const float* __restrict__, const A1Args);
#else
namespace
{
template <int ACT_T, bool WT, bool PERSIST = false>
hipError_t launch_q_inst(const A1Args& a, int n_blocks, hipStream_t stream)
{
  static DynamicLdsLimit lds_limit; // per instantiation, tracked per device (kernels.h)
  const hipError_t e = lds_limit.ensure(reinterpret_cast<const void*>(&nam_a1_q_kernel<ACT_T, WT, PERSIST>), aq::kLdsBytes);
  if (e != hipSuccess)
    return e;
  nam_launch((nam_a1_q_kernel<ACT_T, WT, PERSIST>), dim3(n_blocks), dim3(aq::kNst * 64), aq::kLdsBytes, stream, a.blob, a);
  return hipGetLastError();
}
template <int ACT_T>
hipError_t launch_q_act(const A1Args& a, int n_blocks, hipStream_t stream)
{
  if (a.p_ring) // persistent session (kernel_a1_p4.hip: launch_p4_shape)
    return a.p_out_host != 0 ? launch_q_inst<ACT_T, true, true>(a, n_blocks, stream) : launch_q_inst<ACT_T, false, true>(a, n_blocks, stream);
  const bool wt = a.n_frames <= 2 * kBlock; // short launches write ring appends through
  return wt ? launch_q_inst<ACT_T, true>(a, n_blocks, stream) : launch_q_inst<ACT_T, false>(a, n_blocks, stream);
}
} // namespace

// a.tiles_off: blob offset (floats) of the kernel's own weight block (plan_a1.cpp: build_a1_q; aq_table.h), a.consts_off: of the
// big layers' register tiles = the A1 kernels' tile area (A1Plan::ws_tiles_off, FULL layout)
bool a1_q_takes(int act)
{
  return act == ACT_FASTTANH || act == ACT_TANH;
}
hipError_t launch_a1_q(const A1Args& a, int n_blocks, int act, hipStream_t stream)
{
  if (a.dbg && !a.p_ring) // developer tool (nam_hip_batch_debug_timeline): the stamped instantiation
  {
    static DynamicLdsLimit lds_limit;
    const hipError_t e = lds_limit.ensure(reinterpret_cast<const void*>(&nam_a1_q_kernel<ACT_FASTTANH, false, false, true>), aq::kLdsBytes);
    if (e != hipSuccess)
      return e;
    hipLaunchKernelGGL((nam_a1_q_kernel<ACT_FASTTANH, false, false, true>), dim3(n_blocks), dim3(aq::kNst * 64), aq::kLdsBytes, stream, a.blob, a);
    return hipGetLastError();
  }
  if (act == ACT_FASTTANH)
    return launch_q_act<ACT_FASTTANH>(a, n_blocks, stream);
  if (act == ACT_TANH)
    return launch_q_act<ACT_TANH>(a, n_blocks, stream);
  return hipErrorInvalidValue; // (other activations keep nam_a1_p4_kernel: a1_q_takes)
}
#endif

} // namespace namhip
