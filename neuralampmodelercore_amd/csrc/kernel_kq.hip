// kernel_kq.hip — nam_kq_kernel: the A2 topology (kp_table.h) with ONE LANE PER FRAME, one wavefront per pipeline stage, on
// the 4 x 4 x 1 matrix instruction: no padded matrix rows, no barriers inside a stage.
#include "device_common.h"
#include "il_common.h"
#include "kp_table.h"

namespace namhip
{

// ================================================================================================
// nam_kt_mfma_kernel (kernel_kt_mfma.hip) runs A2-Full — 8 channels, 23 layers of 6 or 15 taps at dilations up to 239, a 16-tap
// head rechannel: the shape of the reference's own fused path, NAM/wavenet/a2_fast.cpp — with one wavefront per SIMD from a
// run-time chunk table: ~340 instructions around every 12 matrix instructions and every latency exposed (35 us per 64-frame
// buffer at 256 streams). At 8 channels rows 8 .. 15 of every 16 x 16 x 4 tile are zero, and an fp32 MFMA and a vector
// instruction never execute at the same time on this chip (SQ_VALU_MFMA_COEXEC_CYCLES = 0): both halves of that cost add up.
// (Round 3's first answer, a pipeline of three four-wave sets on the same tiles — nam_kp_kernel, 13.1 us per buffer — was
// retired in round 5: this kernel took over its activations.) This kernel is compiled for the topology (kp_table.h; plan_a1.cpp:
// build_a1_kp checks a model against it), keeps the K-tap kernel's state (rings, write positions: the two alternate freely on
// one stream) and the session protocol of the A1 pipelines, and changes the shape of the work:
//   * lane t of a wave = frame t of the buffer; the lane holds the layer's 8 input channels (x), the head accumulator and
//     the sums in registers. v_mfma_f32_4x4x1_16b_f32 computes sixteen independent 4 x 4 blocks: block = four frames,
//     A = W[4 h + lane % 4][c] (the same four values in every block), B = the lane's own x[c], D[e] = output channel
//     4 h + e of the lane's frame. A tap is 16 instructions (2 halves x 8 input channels, ~8.6 cycles each), a layer of
//     six taps + 1x1 is 112 against 4 waves x 14 x 32 cycles of the padded 16 x 16 x 4 form: half the matrix cycles, a quarter
//     of the other instructions;
//   * NST = 12 stages of ONE wave each (kq::first_job: jobs cut so that three stages per SIMD balance, kq::stage_of_wave),
//     stage s on buffer n - s, one-slot LDS queues between them (x, head accumulator, input sample, token);
//   * a tap's operand: lookback 0 = the registers; 0 < L <= T = the layer's LDS WINDOW [tail: the last T rows in front
//     of this buffer | the current 64 rows], two planes of 16-byte rows (conflict-free b128 reads at a compile-time
//     offset from the lane's row; T: kq::tail_rows — below one buffer, or the layer's whole history where that is at most
//     192 rows); L > T = rows of the layer's HBM ring, requested ONE BUFFER AHEAD: the registers of a far tap are asked
//     again for the next buffer right behind the instructions that consumed them;
//   * weights: one 256-byte tile per tap in LDS [lane % 4][h][c], read as four broadcast b128.
// Sums: per output half one chain per layer seeded with bias + mixin * input, taps oldest first, input channels in order.
// ================================================================================================
using kq_i4 = __attribute__((ext_vector_type(4))) int;
__device__ mf::f4 kq_sb_load4(kq_i4 rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.v4f32");
__device__ void kq_sb_store4(mf::f4 v, kq_i4 rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.store.v4f32");

namespace kq
{
using namespace kp;
constexpr int kNst = 12;
constexpr int kNoRow = 1 << 26; // a ring row index no descriptor holds: the access is dropped / returns 0
constexpr int kRows = 1 << 20; // num_records of the ring descriptor (rows)
constexpr int kActLeakyMax = 100; // ACT_T of the LeakyReLU instantiation (slope <= 1)
// stage s = jobs [kFirst[s], kFirst[s + 1]); wave i of the workgroup runs stage kStageOfWave[i] — waves i, i + 4, i + 8 share a
// SIMD, and the three stages of every SIMD carry (nearly) the same number of matrix instructions + per-job overhead
// (7 % above the mean for the heaviest; found by exhaustive search over the cuts, tools note in docs/DESIGN_HISTORY.md §4.2d)
constexpr int kFirst[kNst + 1] = {0, 2, 4, 6, 8, 10, 12, 14, 15, 17, 20, 22, 24};
constexpr int kStageOfWave[kNst] = {4, 3, 2, 0, 7, 8, 6, 1, 9, 10, 11, 5};
static_assert(kFirst[kNst] == kJobs, "kq stage table");
constexpr int first_job(int s) { return kFirst[s]; }
constexpr int max_jobs()
{
  int m = 0;
  for (int s = 0; s < kNst; s++)
    m = kFirst[s + 1] - kFirst[s] > m ? kFirst[s + 1] - kFirst[s] : m;
  return m;
}
constexpr int lookback(int job, int j) { return (kKs[job] - 1 - j) * kDs[job]; }
// A job's LDS window holds its input's last T rows in front of the current buffer's 64: taps up to T back read it at a
// compile-time offset from the lane's row, taps further back ("far") read the HBM ring one buffer ahead. T = the largest
// lookback below one buffer — or the job's WHOLE history where that is at most kWinMax rows (three rows per lane: the
// A2 topology's dilation-17 layers, 85 rows, and its 15-tap dilation-13 layer, 182: sixteen far taps and four rings'
// appends per buffer less; the tail then shifts down by 64 rows per buffer instead of being overwritten)
constexpr int kWinMax = 3 * kBlock;
constexpr int win_limit(int job)
{
  const int all = (kKs[job] - 1) * kDs[job];
  return (all >= kBlock && all <= kWinMax) ? all : kBlock - 1;
}
constexpr int tail_rows(int job)
{
  int t = 0;
  for (int j = 0; j < kKs[job]; j++)
  {
    const int L = lookback(job, j);
    t = (L <= win_limit(job) && L > t) ? L : t;
  }
  return t;
}
constexpr bool is_far(int job, int j) { return lookback(job, j) > tail_rows(job); }
constexpr int win_rows(int job) { return tail_rows(job) > 0 ? tail_rows(job) + kBlock + 1 : 0; } // tail | current | one dump row
constexpr int far_taps(int job)
{
  int c = 0;
  for (int j = 0; j < kKs[job]; j++)
    c += is_far(job, j) ? 1 : 0;
  return c;
}
constexpr int far_before(int job, int j) // far taps of the job in front of tap j
{
  int c = 0;
  for (int i = 0; i < j; i++)
    c += is_far(job, i) ? 1 : 0;
  return c;
}
constexpr int stage_far0(int s, int job) // far taps of stage s in front of `job`
{
  int c = 0;
  for (int i = kFirst[s]; i < job; i++)
    c += far_taps(i);
  return c;
}
constexpr int max_far()
{
  int m = 0;
  for (int s = 0; s < kNst; s++)
    m = stage_far0(s, kFirst[s + 1]) > m ? stage_far0(s, kFirst[s + 1]) : m;
  return m;
}
constexpr int tap0(int job) // first tile of a job (tiles: every job's taps in order, then the layers' 1x1)
{
  int t = 0;
  for (int i = 0; i < job; i++)
    t += kKs[i];
  return t;
}
constexpr int kTaps = tap0(kJobs);
constexpr int kTiles = kTaps + kLayers;
constexpr int kTileB = 256; // [lane % 4][h][c]
constexpr int kConstB = 96; // per job: bias[8] | mixin[8] | 1x1 bias[8]
// blob block (floats): tiles | constants | rechannel column (8, padded to 16)
constexpr int kBlobConsts = kTiles * 64, kBlobRech = kBlobConsts + kJobs * 24;
// LDS layout (bytes)
constexpr int kTilesB = 0;
constexpr int kConstsB = kTilesB + kTiles * kTileB;
constexpr int kLdsSrcFloats = kBlobRech; // tiles + constants are copied as they lie
constexpr int kWinB = kConstsB + kJobs * kConstB;
constexpr int win_b(int job)
{
  int b = kWinB;
  for (int i = 0; i < job; i++)
    b += win_rows(i) * 32;
  return b;
}
constexpr int kFlagB = win_b(kJobs); // 256 bytes of single-writer words
constexpr int kQueueB = kFlagB + 256;
constexpr int kSlotB = 4 * 1024 + 256 + 16; // x (two planes), head accumulator (two planes), input sample, token
constexpr int kLdsBytes = kQueueB + (kNst - 1) * kSlotB;
static_assert(kLdsBytes <= 160 * 1024, "kq LDS layout");
static_assert(kConstsB % 16 == 0 && kWinB % 16 == 0 && kQueueB % 16 == 0, "kq LDS alignment");
constexpr int tile_b(int job, int j) { return kTilesB + (tap0(job) + j) * kTileB; }
constexpr int w1_b(int layer) { return kTilesB + (kTaps + layer) * kTileB; }
constexpr int const_b(int job) { return kConstsB + job * kConstB; }
} // namespace kq

template <int ACT_T, bool WT, bool PERSIST>
__global__ __launch_bounds__(kq::kNst * 64) void nam_kq_kernel(const float* __restrict__ blob, const A1Args a)
{
  using namespace mf;
  using il::kOob;
  using i4 = kq_i4;
  constexpr int NST = kq::kNst, NJ = kq::kJobs, MAXJ = kq::max_jobs(), MAXF = kq::max_far();
  extern __shared__ __attribute__((aligned(16))) float lds_kq[];
  char* const lds = reinterpret_cast<char*>(lds_kq);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wall = uni(tid >> 6);
  int S = 0; // this wave's stage
#pragma unroll
  for (int i = 0; i < NST; i++)
    S = wall == i ? kq::kStageOfWave[i] : S;
  const int stream = a.stream_map ? a.stream_map[blockIdx.x] : (int)blockIdx.x;
  float* st = a.state + (size_t)stream * a.state_stride;
  const int n_blocks = PERSIST ? (1 << 30) : (a.n_frames + kBlock - 1) / kBlock;

  const int frame = lane; // this lane's frame inside the buffer
  const float* in = a.in ? a.in + (size_t)stream * a.io_stride : nullptr;
  float* out = a.out ? a.out + (size_t)stream * a.io_stride : nullptr;
  const float head_scale = a.head_scale;
  const float act_p0 = a.act_p0;
  const int act = a.act; // (only read by the run-time-dispatch instantiation)
  const unsigned frame16 = (unsigned)frame * 16u; // the lane's row in a plane of 16-byte rows
  const unsigned cls64 = (unsigned)(lane & 3) * 64u; // its weight record inside a tile
  const int io_bytes = PERSIST ? 0x7ffffff0 : a.n_frames * 4;
  const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc((void*)(in ? in : st), 0, in ? io_bytes : 0, 0x00020000);
  const auto rsrc_out = __builtin_amdgcn_make_buffer_rsrc((void*)(out ? out : st), 0, out ? io_bytes : 0, 0x00020000);
  const unsigned long long in_addr = (unsigned long long)(in ? in : st);
  const i4 in_desc = {uni((int)(unsigned)in_addr), uni((int)(unsigned)(in_addr >> 32) & 0xffff), in ? io_bytes : 0, 0x00020000};
  int* wpos_tbl = reinterpret_cast<int*>(st);
  const int wposv = lane < NJ ? wpos_tbl[lane] : 0; // lane r = write position of ring r (ring r = job r) at launch
  int* const flags = reinterpret_cast<int*>(lds + kq::kFlagB);

  // ---- tiles and constants -> LDS, once per launch, by every wave ----
  constexpr int NT = NST * 64;
  constexpr int kSrc4 = kq::kLdsSrcFloats / 4; // 16-byte records
  constexpr int kS4 = (kSrc4 + NT - 1) / NT;
  {
    const f4* __restrict__ src = reinterpret_cast<const f4*>(blob + a.tiles_off);
#pragma unroll 1
    for (int i0 = 0; i0 < kS4; i0 += 4)
    {
      f4 t[4];
#pragma unroll
      for (int i = 0; i < 4; i++)
        t[i] = src[min((i0 + i) * NT + tid, kSrc4 - 1)];
#pragma unroll
      for (int i = 0; i < 4; i++)
        if ((i0 + i) * NT + tid < kSrc4)
          lds_st4(lds, (unsigned)kq::kTilesB + (unsigned)((i0 + i) * NT + tid) * 16u, t[i]);
    }
  }
  const f4 rech0 = *reinterpret_cast<const f4*>(blob + a.tiles_off + kq::kBlobRech);
  const f4 rech1 = *reinterpret_cast<const f4*>(blob + a.tiles_off + kq::kBlobRech + 4);

  // The stream's rings through ONE descriptor with the row pitch (32 bytes) as the stride: an access names its row by
  // index and its ring by the scalar offset; kNoRow drops it.
  const unsigned long long st_addr = (unsigned long long)st;
  const i4 rs = {uni((int)(unsigned)st_addr), uni((int)((unsigned)(st_addr >> 32) & 0xffffu) | ((kq::kC * 4) << 16)), kq::kRows, 0x00020000};
  auto app_of = [&](int nv) { return frame < nv ? frame : kq::kNoRow; };
  int app_idx = app_of(kBlock); // this lane's frame as a row offset when it appends
  // the write positions of this stage's rings as SCALARS (wp[u] = ring of job J0 + u)
  int wp[MAXJ];
  il::for_each_index(
    [&](auto s_tag) {
      constexpr int SS = decltype(s_tag)::value;
      if (S == SS)
      {
        constexpr int J0 = kq::first_job(SS), NJS = kq::first_job(SS + 1) - J0;
#pragma unroll
        for (int u = 0; u < MAXJ; u++)
          wp[u] = u < NJS ? __builtin_amdgcn_readlane(wposv, J0 + (u < NJS ? u : 0)) : 0;
      }
    },
    std::make_integer_sequence<int, NST>{});
  // a ring row per lane: frame f of the buffer that starts at write position `wpj`, `L` frames back
  struct Row
  {
    f4 q0, q1;
  };
  auto fetch = [&](Row& R_, auto tj_tag, int L, int wpj, int fq) {
    constexpr int TJ = decltype(tj_tag)::value;
    constexpr int RL = kq::ring_len(TJ);
    int sb_ = wpj - L;
    sb_ += sb_ < 0 ? RL : 0;
    const unsigned v = (unsigned)(sb_ + fq);
    const int idx = (int)min(v, v - (unsigned)RL);
    R_.q0 = kq_sb_load4(rs, idx, 0, kq::ring_off(TJ) * 4, 0);
    R_.q1 = kq_sb_load4(rs, idx, 16, kq::ring_off(TJ) * 4, 0);
  };
  Row rows[MAXF > 0 ? MAXF : 1];
  float inp = 0.0f;

  constexpr bool kOutHost = PERSIST && WT; // (kernel_a1_p4.hip: a session whose results go to host memory)
  constexpr int kInAux = PERSIST ? 17 : 0; // session inputs bypass the caches (the caller may rewrite the buffer between commands)
  auto ring_load = [&](unsigned s_) {
    return __hip_atomic_load(a.p_ring + (s_ & (unsigned)a.p_ring_mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  };
  // ---- queues: kernel_a1_p4.hip's one-slot queues, one wave on either side ----
  const unsigned flag_b = (unsigned)kq::kFlagB;
  auto wait_word = [&](unsigned byte_addr, int want) { // until the word has reached `want`
    int tmp;
    asm volatile("1:\n\tds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_sub_u32 %0, %0, %2\n\tv_cmp_gt_i32 vcc, 0, %0\n\t"
                 "s_cbranch_vccz 2f\n\ts_sleep 1\n\ts_branch 1b\n2:"
                 : "=&v"(tmp)
                 : "v"(byte_addr), "v"(want)
                 : "vcc");
  };
  auto queue_put = [&](int q, int k, const f4& x0, const f4& x1, const f4& h0, const f4& h1, float vc, const i4& tok) {
    const unsigned slot = (unsigned)kq::kQueueB + (unsigned)(q * kq::kSlotB);
    const unsigned prod = flag_b + (unsigned)(16 + 2 * q) * 4u, cons = flag_b + (unsigned)(16 + 2 * q + 1) * 4u;
    wait_word(cons, k); // the slot is free once buffer k - 1 has been taken out of it
    asm volatile("" ::: "memory");
    lds_st4(lds, slot + frame16, x0);
    lds_st4(lds, slot + 1024u + frame16, x1);
    lds_st4(lds, slot + 2048u + frame16, h0);
    lds_st4(lds, slot + 3072u + frame16, h1);
    *reinterpret_cast<float*>(lds + slot + 4096u + (unsigned)lane * 4u) = vc;
    if (lane == 0)
      *reinterpret_cast<i4*>(lds + slot + 4352u) = tok;
    asm volatile("" ::: "memory");
    if (lane == 0)
      __hip_atomic_store(reinterpret_cast<int*>(lds + prod), k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
  };
  auto queue_take = [&](int q, int k, f4& x0, f4& x1, f4& h0, f4& h1, float& vc, i4& tok) {
    const unsigned slot = (unsigned)kq::kQueueB + (unsigned)(q * kq::kSlotB);
    const unsigned prod = flag_b + (unsigned)(16 + 2 * q) * 4u, cons = flag_b + (unsigned)(16 + 2 * q + 1) * 4u;
    wait_word(prod, k + 1);
    asm volatile("" ::: "memory");
    tok = *reinterpret_cast<const i4*>(lds + slot + 4352u);
    x0 = lds_ld4(lds, slot + frame16);
    x1 = lds_ld4(lds, slot + 1024u + frame16);
    h0 = lds_ld4(lds, slot + 2048u + frame16);
    h1 = lds_ld4(lds, slot + 3072u + frame16);
    vc = *reinterpret_cast<const float*>(lds + slot + 4096u + (unsigned)lane * 4u);
    asm volatile("" ::"v"(x0), "v"(x1), "v"(h0), "v"(h1), "v"(vc), "v"(tok) : "memory"); // (in registers: the slot may be reused)
    if (lane == 0)
      __hip_atomic_store(reinterpret_cast<int*>(lds + cons), k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
  };

  // ---- per stage, from the state only: the far taps' rows of the first buffer, the tails of the windows ----
  Row tails[MAXJ];
  il::for_each_index(
    [&](auto s_tag) {
      constexpr int SS = decltype(s_tag)::value;
      if (S == SS)
      {
        constexpr int J0 = kq::first_job(SS), NJS = kq::first_job(SS + 1) - J0;
        il::for_each_index(
          [&](auto u_tag) {
            constexpr int U = decltype(u_tag)::value, TJ = J0 + U;
            constexpr int K = kq::kKs[TJ], T = kq::tail_rows(TJ);
            constexpr int F0 = kq::stage_far0(SS, TJ);
            il::for_each_index(
              [&](auto j_tag) {
                constexpr int j = decltype(j_tag)::value;
                constexpr int L = kq::lookback(TJ, j);
                if constexpr (kq::is_far(TJ, j))
                  fetch(rows[F0 + kq::far_before(TJ, j)], std::integral_constant<int, TJ>{}, L, wp[U], frame);
              },
              std::make_integer_sequence<int, K>{});
            if constexpr (T > 0 && T <= kBlock) // tail row r = the frame T - r before the first buffer: lanes r < T
              fetch(tails[U], std::integral_constant<int, TJ>{}, T, wp[U], frame < T ? frame : kq::kNoRow);
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
  if (S == 0)
    inp = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_in, frame * 4, uni((int)boff0), kInAux));
  // the tails of this stage's windows (lanes r >= T write the dump row)
  il::for_each_index(
    [&](auto s_tag) {
      constexpr int SS = decltype(s_tag)::value;
      if (S == SS)
      {
        constexpr int J0 = kq::first_job(SS), NJS = kq::first_job(SS + 1) - J0;
        il::for_each_index(
          [&](auto u_tag) {
            constexpr int U = decltype(u_tag)::value, TJ = J0 + U;
            constexpr int T = kq::tail_rows(TJ), WR = kq::win_rows(TJ), WB0 = kq::win_b(TJ), WB1 = WB0 + WR * 16;
            if constexpr (T > 0 && T <= kBlock)
            {
              const unsigned row = (unsigned)(frame < T ? frame : T + kBlock) * 16u;
              lds_st4(lds, (unsigned)WB0 + row, tails[U].q0);
              lds_st4(lds, (unsigned)WB1 + row, tails[U].q1);
            }
            else if constexpr (T > kBlock)
            {
              // the job's whole history lives in LDS: the ring as it lies in the state, row for row
              constexpr int RL = kq::ring_len(TJ);
#pragma unroll
              for (int r0 = 0; r0 < RL; r0 += kBlock)
              {
                const int r = r0 + frame;
                const int idx = r < RL ? r : kq::kNoRow;
                const f4 q0 = kq_sb_load4(rs, idx, 0, kq::ring_off(TJ) * 4, 0), q1 = kq_sb_load4(rs, idx, 16, kq::ring_off(TJ) * 4, 0);
                const unsigned rw = (unsigned)(r < RL ? r : RL) * 16u;
                lds_st4(lds, (unsigned)WB0 + rw, q0);
                lds_st4(lds, (unsigned)WB1 + rw, q1);
              }
            }
          },
          std::make_integer_sequence<int, NJS>{});
      }
    },
    std::make_integer_sequence<int, NST>{});
  lds_barrier(); // weights, constants and flags are in place

  f4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = x0, head0 = x0, head1 = x0;
  int nvalid = kBlock; // frames of the buffer this wave's stage is working on
  // every buffer of this launch is whole (a session; a launch of a multiple of 64 frames): layers without far taps keep their
  // history in the LDS window only and write its tail back when the launch leaves
  const bool lazy = PERSIST || (a.n_frames % kBlock) == 0;
  float cond = 0.0f;
  unsigned long long spec_cmd = 0; // PERSIST, stage 0: the early look at the next command ...
  float inp_spec = 0.0f; // ... and the input sample requested on a hit
  bool more = false; // this stage has another buffer behind the current one
  unsigned boff = 0; // byte offset of the current buffer in the stream's row

  // One job = one layer (or the head rechannel), everything about it known at compile time. model.cpp:183-393:
  // z = act(conv(x) + mixin(cond)); head += z; x += layer1x1(z); model.cpp:513-531 for the head rechannel.
  auto job = [&](auto j_tag, auto s_tag) {
    constexpr int JI = decltype(j_tag)::value;
    constexpr int SS = decltype(s_tag)::value;
    constexpr int J0 = kq::first_job(SS);
    constexpr int U = JI - J0; // position in the stage
    constexpr bool HEAD = JI == kq::kLayers;
    constexpr int K = kq::kKs[JI], RL = kq::ring_len(JI);
    constexpr int T = kq::tail_rows(JI), WR = kq::win_rows(JI), WB0 = kq::win_b(JI), WB1 = WB0 + WR * 16;
    constexpr int F0 = kq::stage_far0(SS, JI);
    __builtin_amdgcn_sched_barrier(0);
    const f4 in0 = HEAD ? head0 : x0, in1 = HEAD ? head1 : x1;
    // (a) the job's input -> its history ring: row (position + frame) mod R; -> the current rows of its window. A layer whose
    // taps all lie inside one buffer (its window's tail IS its history: eleven of the A2 topology's 24 jobs) appends
    // nothing while whole buffers flow — the tail goes back into the ring once, when the launch leaves (below)
    const int wpj = wp[U];
    if (!(lazy && kq::far_taps(JI) == 0 && T > 0))
    {
      const unsigned v = (unsigned)(wpj + app_idx);
      const int widx = (int)min(v, v - (unsigned)RL);
      kq_sb_store4(in0, rs, widx, 0, kq::ring_off(JI) * 4, WT && !PERSIST ? 17 : 0);
      kq_sb_store4(in1, rs, widx, 16, kq::ring_off(JI) * 4, WT && !PERSIST ? 17 : 0);
    }
    unsigned old16 = 0; // resident ring: byte offset of the lane's OLDEST tap row, (position - T + frame) mod R; tap j is j d rows on
    if constexpr (T > kBlock)
    {
      int sb_ = wpj - T;
      sb_ += sb_ < 0 ? RL : 0;
      const unsigned v = (unsigned)(sb_ + frame);
      old16 = min(v, v - (unsigned)RL) * 16u;
      // this buffer's row: (position + frame) mod R, as the append to the HBM ring would have it = T rows on from the oldest tap
      const unsigned c = old16 + (unsigned)(T * 16);
      const unsigned cur16 = min(c, c - (unsigned)(RL * 16));
      lds_st4(lds, (unsigned)WB0 + cur16, in0);
      lds_st4(lds, (unsigned)WB1 + cur16, in1);
      asm volatile("" ::: "memory");
    }
    else if constexpr (T > 0)
    {
      lds_st4(lds, (unsigned)(WB0 + T * 16) + frame16, in0);
      lds_st4(lds, (unsigned)(WB1 + T * 16) + frame16, in1);
      // the taps read OTHER lanes' rows: the compiler, which reasons per lane, must not move those reads above these
      // stores (the LDS itself runs a wave's accesses in order)
      asm volatile("" ::: "memory");
    }
    int wpn = wpj + nvalid; // the ring's position for the next buffer (scalar unit)
    wpn -= wpn >= RL ? RL : 0;
    // (b) constants
    f4 acc0 = lds_ld4(lds, (unsigned)kq::const_b(JI)), acc1 = lds_ld4(lds, (unsigned)kq::const_b(JI) + 16u);
    if constexpr (!HEAD)
    {
      const f4 m0 = lds_ld4(lds, (unsigned)kq::const_b(JI) + 32u), m1 = lds_ld4(lds, (unsigned)kq::const_b(JI) + 48u);
      const f4 c4 = {cond, cond, cond, cond};
      acc0 = __builtin_elementwise_fma(m0, c4, acc0);
      acc1 = __builtin_elementwise_fma(m1, c4, acc1);
    }
    // persistent session, stage 0: look at the next ring slot in job 0 and, when the command is already there, request
    // the next buffer's input sample from it in job 1 (unconditional load, out-of-range offset on a miss)
    if constexpr (PERSIST && JI == 0)
    {
      const unsigned long long v = ring_load(na + 1); // (lane 0's view for the whole wave, as at the end of the buffer)
      spec_cmd = ((unsigned long long)(unsigned)uni((int)(unsigned)(v >> 32)) << 32) | (unsigned long long)(unsigned)uni((int)(unsigned)v);
    }
    if constexpr (PERSIST && JI == 1)
    {
      const bool hit = (unsigned)(spec_cmd >> 32) == na + 2;
      const int soff = uni(hit ? (int)((unsigned)spec_cmd * 4u) : 0);
      inp_spec = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_in, hit ? frame * 4 : (int)kOob, soff, kInAux));
    }
    // (c) the taps, oldest first
    il::for_each_index(
      [&](auto t_tag) {
        constexpr int j = decltype(t_tag)::value;
        constexpr int L = kq::lookback(JI, j);
        constexpr unsigned tb = (unsigned)kq::tile_b(JI, j);
        const f4 wa = lds_ld4(lds, tb + cls64), wb = lds_ld4(lds, tb + 16u + cls64);
        f4 b0, b1;
        if constexpr (L == 0)
          b0 = in0, b1 = in1;
        else if constexpr (kq::is_far(JI, j))
          b0 = rows[F0 + kq::far_before(JI, j)].q0, b1 = rows[F0 + kq::far_before(JI, j)].q1;
        else if constexpr (T > kBlock)
        {
          // (T - L) rows on from the oldest tap's row, wrapped once: three vector instructions, no scalar ones
          const unsigned c = old16 + (unsigned)((T - L) * 16);
          const unsigned a16 = L == T ? old16 : min(c, c - (unsigned)(RL * 16));
          b0 = lds_ld4(lds, (unsigned)WB0 + a16);
          b1 = lds_ld4(lds, (unsigned)WB1 + a16);
        }
        else
        {
          b0 = lds_ld4(lds, (unsigned)(WB0 + (T - L) * 16) + frame16);
          b1 = lds_ld4(lds, (unsigned)(WB1 + (T - L) * 16) + frame16);
        }
        if constexpr (HEAD)
        {
#pragma unroll
          for (int c = 0; c < 4; c++)
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wa[c], b0[c], acc0, 0, 0, 0);
#pragma unroll
          for (int c = 0; c < 4; c++)
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wb[c], b1[c], acc0, 0, 0, 0);
        }
        else
        {
          const f4 wc = lds_ld4(lds, tb + 32u + cls64), wd = lds_ld4(lds, tb + 48u + cls64);
#pragma unroll
          for (int c = 0; c < 4; c++)
          {
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wa[c], b0[c], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wc[c], b0[c], acc1, 0, 0, 0);
          }
#pragma unroll
          for (int c = 0; c < 4; c++)
          {
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wb[c], b1[c], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wd[c], b1[c], acc1, 0, 0, 0);
          }
        }
        // a far tap's rows for the NEXT buffer, into the registers just consumed (rows up to this buffer's last frame: this
        // wave appended them at the top of the job)
        if constexpr (kq::is_far(JI, j))
          fetch(rows[F0 + kq::far_before(JI, j)], j_tag, L, wpn, frame);
      },
      std::make_integer_sequence<int, K>{});
    // (d) the window's tail for the next buffer: the last T rows of this input (behind the taps' reads: LDS runs in order)
    if constexpr (T > 0 && T <= kBlock)
    {
      asm volatile("" ::: "memory"); // (behind every tap's read of the old tail)
      const unsigned row = (unsigned)(frame >= kBlock - T ? frame - (kBlock - T) : T + kBlock) * 16u;
      lds_st4(lds, (unsigned)WB0 + row, in0);
      lds_st4(lds, (unsigned)WB1 + row, in1);
      asm volatile("" ::: "memory");
    }
    // (e) epilogue
    if constexpr (HEAD)
    {
      const float yout = head_scale * acc0[0];
      if (kOutHost && a.p_out_host == 2) // ticketed host buffers: written through (kernel_a1_q.hip)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yout), rsrc_out, frame < nvalid ? frame * 4 : (int)kOob, uni((int)boff), 17);
      else
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yout), rsrc_out, frame < nvalid ? frame * 4 : (int)kOob, uni((int)boff),
                                              PERSIST && !kOutHost ? 17 : 0);
    }
    else
    {
      const f4 b1v0 = lds_ld4(lds, (unsigned)kq::const_b(JI) + 64u), b1v1 = lds_ld4(lds, (unsigned)kq::const_b(JI) + 80u);
      constexpr unsigned tb = (unsigned)kq::w1_b(JI);
      const f4 wa = lds_ld4(lds, tb + cls64), wb = lds_ld4(lds, tb + 16u + cls64);
      const f4 wc = lds_ld4(lds, tb + 32u + cls64), wd = lds_ld4(lds, tb + 48u + cls64);
      f4 z0, z1;
      if constexpr (ACT_T == kq::kActLeakyMax) // LeakyReLU with a slope <= 1: max(v, slope * v) (launch_kq checks the slope)
      {
        z0 = __builtin_elementwise_max(acc0, acc0 * act_p0);
        z1 = __builtin_elementwise_max(acc1, acc1 * act_p0);
      }
      else
      {
        z0 = act4<ACT_T>(act, acc0, act_p0);
        z1 = act4<ACT_T>(act, acc1, act_p0);
      }
      head0 += z0;
      head1 += z1;
      f4 y0 = x0 + b1v0, y1 = x1 + b1v1;
#pragma unroll
      for (int c = 0; c < 4; c++)
      {
        y0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wa[c], z0[c], y0, 0, 0, 0);
        y1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wc[c], z0[c], y1, 0, 0, 0);
      }
#pragma unroll
      for (int c = 0; c < 4; c++)
      {
        y0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wb[c], z1[c], y0, 0, 0, 0);
        y1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wd[c], z1[c], y1, 0, 0, 0);
      }
      x0 = y0;
      x1 = y1;
    }
    wp[U] = wpn;
  };

  // ---- the stage loops (one wave per stage: its decisions are its own) ----
  auto run = [&](auto s_tag) {
    constexpr int SS = decltype(s_tag)::value;
    constexpr int J0 = kq::first_job(SS), NJS = kq::first_job(SS + 1) - J0;
    constexpr bool FIRST = SS == 0, LAST = SS == NST - 1;
    constexpr int QIN = SS - 1, QOUT = SS;
    boff = boff0;
    bool have = !FIRST || n_blocks > 0;
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
        i4 tok;
        queue_take(QIN, k, x0, x1, head0, head1, cond, tok);
        boff = (unsigned)uni(tok[0]);
        nvalid = uni(tok[1]);
        exit_tok = uni(tok[2]) != 0;
        more = PERSIST || uni(tok[3]) != 0;
      }
      auto hand_over = [&](bool is_exit) {
        queue_put(QOUT < 0 ? 0 : QOUT, k, x0, x1, head0, head1, cond, i4{(int)boff, nvalid, is_exit ? 1 : 0, more ? 1 : 0});
      };
      if (exit_tok)
      {
        if constexpr (!LAST)
          hand_over(true);
        break;
      }
      if constexpr (!PERSIST)
      {
        if (nvalid != kBlock) // a ragged last block: only its frames are appended
          app_idx = app_of(nvalid);
      }
      if constexpr (FIRST)
      {
        cond = inp; // this buffer's input sample (requested a buffer ago)
        if constexpr (!PERSIST) // next block's (offset beyond the launch's frames -> 0)
          inp = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_in, frame * 4, uni((k + 1) * (kBlock * 4)), 0));
        x0 = rech0 * cond;
        x1 = rech1 * cond;
        head0 = head1 = f4{0.f, 0.f, 0.f, 0.f};
      }
      il::for_each_index([&](auto u_tag) { job(std::integral_constant<int, J0 + decltype(u_tag)::value>{}, s_tag); },
                         std::make_integer_sequence<int, NJS>{});
      if constexpr (!LAST)
        hand_over(false);
      else if constexpr (PERSIST)
      {
        done++;
        if (kOutHost && a.p_out_host == 2)
        {
          // ... and the workgroup counts itself in behind them (kernel_a1_q.hip; kernels.h: p_cmd_count / p_cmd_done)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
          if (lane == 0 && (done & 15u) == 0u)
            __hip_atomic_store(a.p_prog + blockIdx.x, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        else if (lane == 0 && (done & 15u) == 0u) // progress for the host's ring bookkeeping (not a completion signal)
          __hip_atomic_store(a.p_prog + blockIdx.x, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      if constexpr (FIRST)
      {
        // the next buffer of this stage
        if constexpr (PERSIST)
        {
          const unsigned tag = na + 2u; // the command behind the one just finished
          unsigned long long v = spec_cmd;
          if ((unsigned)(v >> 32) != tag)
          {
            // the early look missed: look again; while the later stages still work, for a microsecond (bounded: the launch
            // never waits for a command)
            v = session_wait_command<16>(a, ring_load, tag, ring_load(tag - 1u), (long long)(a.p_linger > 0 ? a.p_linger : 100)); // (1 us unless the launch lingers)
          }
          // ONE view of the ring slot for the whole wave (lane 0's): the lanes' loads are separate memory requests, and a
          // command that lands between them would split the wave — some lanes leaving, some starting the next buffer
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
              // (the early look came too early) load now and wait right here, inside the asm
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
  };
  il::for_each_index(
    [&](auto s_tag) {
      if (S == decltype(s_tag)::value)
        run(s_tag);
    },
    std::make_integer_sequence<int, NST>{});

  // the tails of the layers that appended nothing (above): window rows [0, T) = the last T frames of the layer's input, into
  // the ring rows in front of its (final) write position — the state is again what every kernel of the family expects
  if (lazy)
  {
    asm volatile("" ::: "memory");
    il::for_each_index(
      [&](auto s_tag) {
        constexpr int SS = decltype(s_tag)::value;
        if (S == SS)
        {
          constexpr int J0 = kq::first_job(SS), NJS = kq::first_job(SS + 1) - J0;
          il::for_each_index(
            [&](auto u_tag) {
              constexpr int U = decltype(u_tag)::value, TJ = J0 + U;
              constexpr int T = kq::tail_rows(TJ), WR = kq::win_rows(TJ), WB0 = kq::win_b(TJ), WB1 = WB0 + WR * 16, RL = kq::ring_len(TJ);
              if constexpr (kq::far_taps(TJ) == 0 && T > 0 && T <= kBlock)
              {
                int fr = frame; // (opaque: see below)
                asm volatile("" : "+v"(fr));
                const unsigned row = (unsigned)(fr < T ? fr : 0) * 16u;
                const f4 q0 = lds_ld4(lds, (unsigned)WB0 + row), q1 = lds_ld4(lds, (unsigned)WB1 + row);
                int sb_ = wp[U] - T;
                sb_ += sb_ < 0 ? RL : 0;
                const unsigned v = (unsigned)(sb_ + fr);
                const int idx = fr < T ? (int)min(v, v - (unsigned)RL) : kq::kNoRow;
                kq_sb_store4(q0, rs, idx, 0, kq::ring_off(TJ) * 4, 0);
                kq_sb_store4(q1, rs, idx, 16, kq::ring_off(TJ) * 4, 0);
              }
              else if constexpr (T > kBlock) // the resident ring goes back as it lies
              {
                int fr = frame; // (opaque: the rows' indices are computed HERE, not hoisted above the stage's loop and kept in registers across it)
                asm volatile("" : "+v"(fr));
#pragma unroll
                for (int r0 = 0; r0 < RL; r0 += kBlock)
                {
                  const int r = r0 + fr;
                  const unsigned row = (unsigned)(r < RL ? r : 0) * 16u;
                  const f4 q0 = lds_ld4(lds, (unsigned)WB0 + row), q1 = lds_ld4(lds, (unsigned)WB1 + row);
                  kq_sb_store4(q0, rs, r < RL ? r : kq::kNoRow, 0, kq::ring_off(TJ) * 4, 0);
                  kq_sb_store4(q1, rs, r < RL ? r : kq::kNoRow, 16, kq::ring_off(TJ) * 4, 0);
                }
              }
            },
            std::make_integer_sequence<int, NJS>{});
        }
      },
      std::make_integer_sequence<int, NST>{});
  }
  // the write positions of this stage's rings go back into the state (lane r = ring r)
  il::for_each_index(
    [&](auto s_tag) {
      constexpr int SS = decltype(s_tag)::value;
      if (S == SS)
      {
        constexpr int J0 = kq::first_job(SS), NJS = kq::first_job(SS + 1) - J0;
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

namespace
{
template <int ACT_T, bool WT, bool PERSIST = false>
hipError_t launch_kq_inst(const A1Args& a, int n_blocks, hipStream_t stream)
{
  static DynamicLdsLimit lds_limit; // per instantiation, tracked per device (kernels.h)
  const hipError_t e = lds_limit.ensure(reinterpret_cast<const void*>(&nam_kq_kernel<ACT_T, WT, PERSIST>), kq::kLdsBytes);
  if (e != hipSuccess)
    return e;
  nam_launch((nam_kq_kernel<ACT_T, WT, PERSIST>), dim3(n_blocks), dim3(kq::kNst * 64), kq::kLdsBytes, stream, a.blob, a);
  return hipGetLastError();
}
template <int ACT_T>
hipError_t launch_kq_act(const A1Args& a, int n_blocks, hipStream_t stream)
{
  if (a.p_ring) // persistent session (kernel_a1_p4.hip: launch_p4_shape)
    return a.p_out_host != 0 ? launch_kq_inst<ACT_T, true, true>(a, n_blocks, stream) : launch_kq_inst<ACT_T, false, true>(a, n_blocks, stream);
  const bool wt = a.n_frames <= 2 * kBlock; // short launches write ring appends through
  return wt ? launch_kq_inst<ACT_T, true>(a, n_blocks, stream) : launch_kq_inst<ACT_T, false>(a, n_blocks, stream);
}
} // namespace

// a.tiles_off: blob offset (floats) of the kernel's weight block (plan_a1.cpp: build_a1_kp — tiles | constants | rechannel column).
// Instantiated: the A2 activation (LeakyReLU with a slope <= 1, as max(v, slope v) — ReLU is its slope 0), Tanh and Fasttanh
// (NAM/activations.h:59-98). The run-time-dispatch form does not fit 168 registers: any other activation on this topology runs
// nam_kt_mfma_kernel, one launch per buffer (the reference's own fused path takes LeakyReLU(0.01) only: a2_fast.cpp:617).
bool kq_takes(int act, float act_p0)
{
  return (act == ACT_LEAKYRELU && act_p0 <= 1.0f) || act == ACT_RELU || act == ACT_TANH || act == ACT_FASTTANH;
}
hipError_t launch_kq(const A1Args& a, int n_blocks, int act, hipStream_t stream)
{
  if (!kq_takes(act, a.act_p0))
    return hipErrorInvalidValue;
  if (act == ACT_TANH)
    return launch_kq_act<ACT_TANH>(a, n_blocks, stream);
  if (act == ACT_FASTTANH)
    return launch_kq_act<ACT_FASTTANH>(a, n_blocks, stream);
  if (act == ACT_RELU)
  {
    A1Args r = a;
    // max(v, 0 v): -0 where ReLU gives +0 — equal in every sum and product behind it — for every FINITE v. Non-finite values
    // differ from the reference's `x > 0 ? x : 0` (NAM/activations.h:59-66): v = -inf gives 0 * v = NaN and the max returns -inf
    // (reference: 0); v = NaN stays NaN (reference: 0). A model whose pre-activations are non-finite has left the range every
    // parity statement of this repository is made for; the finite range is bit-for-bit the LeakyReLU path.
    r.act_p0 = 0.0f;
    return launch_kq_act<kq::kActLeakyMax>(r, n_blocks, stream);
  }
  return launch_kq_act<kq::kActLeakyMax>(a, n_blocks, stream);
}

} // namespace namhip
