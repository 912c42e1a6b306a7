// EXPERIMENT, NOT BUILT (round 3). kernel_kp2.hip — nam_kp2_kernel: nam_kp_kernel (csrc/kernel_kp.hip; read its header first)
// without the matrix padding and without most of its ring requests. Measured on MI355X, A2-Full, 256 streams: parity green
// (every A2 test), matrix-pipe cycles halved (6.2 k per SIMD and buffer), but the vector instructions it adds (the kh combines,
// selects, SGPR spills) ate the gain: 9.9 us per buffer against nam_kp_kernel's 10.4 (+4 %). Restarting sessions then hit an
// intermittent GPU memory access fault (tools/persist_soak.py: 5 of 7 runs; nam_kp_kernel: 0 of 8) that was not found in the
// time left, so it does not ship. To build it: csrc/Makefile KERNELS += this file, plan.cpp gets kp2_plan_packing.inc, A1Plan two
// offsets, launch_kp2 declared in kernels.h.
#include "device_common.h"
#include "il_common.h"
#include "kp_table.h"

namespace namhip
{

// ================================================================================================
// Two facts about nam_kp_kernel's 10.4 us per buffer (A2-Full, 256 streams; profiles/r03/counters_kp_and_p4_resident.txt):
//   * an fp32 MFMA and a vector-ALU instruction never execute at the same time on this chip (SQ_VALU_MFMA_COEXEC_CYCLES =
//     0: the fp32 matrix rate IS the packed-fp32 vector rate, the same FMA lanes) — a step costs the SUM of the two;
//   * at 8 channels half of every 16x16x4 tile is padding: 6.2 k of the 12.4 k matrix cycles per step are wasted, and
//     two thirds of the 6.9 k vector cycles feed ring requests whose rows mostly lie inside the previous buffer.
// So, same pipeline of wave sets (three stages of four waves, wave w = frames 16 w .. 16 w + 15, one-slot LDS queues
// between the stages, "publish, stage barrier" inside them), same state (rings, write positions: the kernels alternate
// freely on one stream), but:
//   * v_mfma_f32_4x4x1_16b_f32 — sixteen independent 4 x 4 blocks, no padding: lane 16 g + n, g = 2 kh + og, computes
//     output channels 4 og .. 4 og + 3 of frame n from input channels 4 kh .. 4 kh + 3 (block = (g, n / 4), A = one
//     weight per lane, B = one input channel of the lane's own frame). A tap is FOUR instructions (8 cycles each) on ONE
//     16-byte operand read instead of two 32-cycle ones; the two kh halves of a sum meet through v_permlane32_swap;
//   * every layer keeps the last rows of its previous input ("tail": the largest lookback below 64, two parities) in
//     LDS: a tap that reaches back less than a buffer reads the published rows or the tail — one compare, one select,
//     one ds_read_b128 — and only lookbacks >= 64 are requested from the HBM ring (two jobs ahead, by row index). The
//     tails of a launch's first buffer come from the rings in the prologue;
//   * tap tiles shrink to 256 bytes (16 lane classes x 4 input channels): 44 KB for the whole model.
// Sums: per layer one chain per kh half seeded (kh = 0) with bias + mixin * input, the halves added at the end; the 1x1
// the same with x + bias. Equal to nam_kt_mfma_kernel / the oracle to ~1e-6.
// ================================================================================================
using kp2_i4 = __attribute__((ext_vector_type(4))) int;
__device__ mf::f4 kp2_sb_load4(kp2_i4 rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.v4f32");
__device__ void kp2_sb_store4(mf::f4 v, kp2_i4 rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.store.v4f32");

namespace kp2
{
using namespace kp;
constexpr int kNoRow = 1 << 26; // a ring row index no descriptor holds: the access is dropped / returns 0
constexpr int kRows = 1 << 20; // num_records of the ring descriptor (rows)
constexpr int kAhead = 2; // a job's ring rows are requested this many jobs earlier (of the same stage, wrapping to the next buffer)
constexpr int kActLeakyMax = 100; // ACT_T of the LeakyReLU instantiation (slope <= 1)
constexpr int kRowB = (kC + 4) * 4; // LDS pitch of a frame row (published buffers, tails, scratch)
constexpr int kBufB = kBlock * kRowB; // one published buffer: row t = frame t
constexpr int tap0(int job) // first tap (tile) of a job
{
  int t = 0;
  for (int i = 0; i < job; i++)
    t += kKs[i];
  return t;
}
constexpr int kTaps = tap0(kJobs);
constexpr int tail_rows(int job) // the largest lookback of the job below one buffer: that many rows of its previous input stay in LDS
{
  int t = 0;
  for (int j = 0; j < kKs[job]; j++)
  {
    const int L = (kKs[job] - 1 - j) * kDs[job];
    t = (L < kBlock && L > t) ? L : t;
  }
  return t;
}
constexpr int tail_row0(int job) // first row of the job's tails (two parities each) in the tail area
{
  int r = 0;
  for (int i = 0; i < job; i++)
    r += 2 * tail_rows(i);
  return r;
}
constexpr int hist_taps(int job) // taps whose operand comes from the HBM ring (lookback >= one buffer)
{
  int c = 0;
  for (int j = 0; j < kKs[job]; j++)
    c += (kKs[job] - 1 - j) * kDs[job] >= kBlock ? 1 : 0;
  return c;
}
constexpr int max_hist()
{
  int m = 1;
  for (int i = 0; i < kJobs; i++)
    m = hist_taps(i) > m ? hist_taps(i) : m;
  return m;
}
// LDS layout (bytes); the first three areas are one contiguous copy of the blob region plan.cpp: build_a1_kp lays down
constexpr int kTilesB = 0; // tap tiles [tap][16 lane classes (g, n & 3)][4 input channels]
constexpr int kW1B = kTilesB + kTaps * 256; // 1x1 tiles [layer][16][4]
constexpr int kConstB = kW1B + kLayers * 256; // constants [job][bias | mixin | 1x1 bias][g][4]: zero for the kh = 1 lanes
constexpr int kWeightFloats = (kTaps + kLayers) * 64 + kJobs * 48;
constexpr int kPubB = kConstB + kJobs * 192; // per stage three published buffers: E (the stage's input), 0, 1 (alternating)
constexpr int pub_b(int stage, int buf) { return kPubB + (stage * 3 + buf) * kBufB; }
constexpr int tail_area_b(int nst) { return kPubB + nst * 3 * kBufB; }
constexpr int tail_b(int nst, int job, int par) { return tail_area_b(nst) + (tail_row0(job) + par * tail_rows(job)) * kRowB; }
constexpr int scratch_b(int nst) { return tail_area_b(nst) + tail_row0(kJobs) * kRowB; } // per wave 16 rows: the activation on its way to the 1x1
constexpr int flag_b(int nst) { return scratch_b(nst) + nst * 4 * 16 * kRowB; } // 256 bytes of single-writer words (kernel_a1_p4.hip)
constexpr int queue_b(int nst) { return flag_b(nst) + 256; }
constexpr int kSlotB = 64 * 16 + 64 * 16 + 64 * 4 + 16;
constexpr int lds_bytes(int nst) { return queue_b(nst) + (nst - 1) * 4 * kSlotB; }
static_assert(lds_bytes(3) <= 160 * 1024, "kp2 LDS layout");
static_assert(kPubB % 16 == 0 && kPubB >= kBlock * kRowB, "kp2 LDS alignment / the operand addresses stay positive");
} // namespace kp2

template <int ACT_T, bool WT, bool PERSIST, int NST>
__global__ __launch_bounds__(NST * 256) void nam_kp2_kernel(const float* __restrict__ blob, const A1Args a)
{
  using namespace mf;
  using il::kOob;
  using i4 = kp2_i4;
  constexpr int NJ = kp2::kJobs, MAXJ = kp2::max_jobs(NST), MAXH = kp2::max_hist();
  extern __shared__ __attribute__((aligned(16))) float lds_kp2[];
  char* const lds = reinterpret_cast<char*>(lds_kp2);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wall = uni(tid >> 6);
  const int S = wall >> 2; // stage
  const int w = wall & 3;
  const int stream = a.stream_map ? a.stream_map[blockIdx.x] : (int)blockIdx.x;
  float* st = a.state + (size_t)stream * a.state_stride;
  const int n_blocks = PERSIST ? (1 << 30) : (a.n_frames + kBlock - 1) / kBlock;

  const int g = lane >> 4; // = 2 kh + og
  const int nn = lane & 15;
  const int frame = 16 * w + nn; // this lane's frame inside the buffer
  const float* in = a.in ? a.in + (size_t)stream * a.io_stride : nullptr;
  float* out = a.out ? a.out + (size_t)stream * a.io_stride : nullptr;
  const float head_scale = a.head_scale;
  const float act_p0 = a.act_p0;
  const int act = a.act; // (only read by the run-time-dispatch instantiation)
  const bool pub_lane = g < 2; // the kh = 0 lanes publish / append: lane (og, n) holds channels 4 og .. 4 og + 3 of frame n
  const unsigned quad_b = (unsigned)(g & 1) * 16u; // the quad a kh = 0 lane publishes
  const unsigned opnd_b = (unsigned)(g >> 1) * 16u; // the lane's B-operand slice of a frame row: channels 4 kh .. 4 kh + 3
  const unsigned cls16 = (unsigned)(g * 4 + (lane & 3)) * 16u; // its record in a tile
  const unsigned lane16 = (unsigned)lane * 16u;
  const int io_bytes = PERSIST ? 0x7ffffff0 : a.n_frames * 4;
  const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc((void*)(in ? in : st), 0, in ? io_bytes : 0, 0x00020000);
  const auto rsrc_out = __builtin_amdgcn_make_buffer_rsrc((void*)(out ? out : st), 0, out ? io_bytes : 0, 0x00020000);
  const unsigned long long in_addr = (unsigned long long)(in ? in : st);
  const i4 in_desc = {uni((int)(unsigned)in_addr), uni((int)(unsigned)(in_addr >> 32) & 0xffff), in ? io_bytes : 0, 0x00020000};
  int* wpos_tbl = reinterpret_cast<int*>(st);
  const int wposv = lane < NJ ? wpos_tbl[lane] : 0; // lane r = write position of ring r (ring r = job r) at launch
  constexpr int kFlagB = kp2::flag_b(NST), kQueueB = kp2::queue_b(NST);
  int* const flags = reinterpret_cast<int*>(lds + kFlagB);

  // ---- tap tiles, 1x1 tiles and constants -> LDS, once per launch, by every wave ----
  constexpr int NT = NST * 256;
  constexpr int kW4 = kp2::kWeightFloats / 4; // 16-byte records
  constexpr int kT4 = (kW4 + NT - 1) / NT;
  f4 tl4[kT4];
  {
    const f4* __restrict__ tsrc = reinterpret_cast<const f4*>(blob + a.tiles_off);
#pragma unroll
    for (int i = 0; i < kT4; i++)
      tl4[i] = tsrc[min(i * NT + tid, kW4 - 1)];
  }
  const f4 rech = *reinterpret_cast<const f4*>(blob + a.r1_off + g * 4);

  // the stream's rings through ONE descriptor with the row pitch (32 bytes) as the stride (kernel_kp.hip)
  const unsigned long long st_addr = (unsigned long long)st;
  const i4 rs = {uni((int)(unsigned)st_addr), uni((int)((unsigned)(st_addr >> 32) & 0xffffu) | ((kp2::kC * 4) << 16)), kp2::kRows, 0x00020000};
  auto app_of = [&](int nv) { return (pub_lane && frame < nv) ? frame : kp2::kNoRow; };
  int app_idx = app_of(kBlock);
  int wp[MAXJ]; // the write positions of this stage's rings as SCALARS
  il::for_each_index(
    [&](auto s_tag) {
      constexpr int SS = decltype(s_tag)::value;
      if (S == SS)
      {
        constexpr int J0 = kp2::first_job(NST, SS), NJS = kp2::first_job(NST, SS + 1) - J0;
#pragma unroll
        for (int u = 0; u < MAXJ; u++)
          wp[u] = u < NJS ? __builtin_amdgcn_readlane(wposv, J0 + (u < NJS ? u : 0)) : 0;
      }
    },
    std::make_integer_sequence<int, NST>{});
  // the ring operands of job TJ — its taps that reach back a buffer or more — for the buffer that starts at write position
  // `wpj` of its ring; `valid`: wave-uniform
  struct Rows
  {
    f4 r[MAXH];
  };
  auto fetch = [&](Rows& R_, auto tj_tag, bool valid, int wpj) {
    constexpr int TJ = decltype(tj_tag)::value;
    constexpr int K = kp2::kKs[TJ], D = kp2::kDs[TJ], RL = kp2::ring_len(TJ), RO = kp2::ring_off(TJ) * 4;
    int hj = 0;
#pragma unroll
    for (int j = 0; j < K - 1; j++)
    {
      const int L = (K - 1 - j) * D;
      if (L >= kBlock)
      {
        int sb_ = wpj - L;
        sb_ += sb_ < 0 ? RL : 0;
        sb_ = valid ? sb_ : kp2::kNoRow;
        const unsigned v = (unsigned)(sb_ + frame);
        const int idx = (int)min(v, v - (unsigned)RL);
        R_.r[hj++] = kp2_sb_load4(rs, idx, (int)opnd_b, RO, 0);
      }
    }
  };
  Rows rows[MAXJ];
  float inp = 0.0f;

  constexpr bool kOutHost = PERSIST && WT; // (kernel_a1_p4.hip: a session whose results go to host memory)
  constexpr int kInAux = PERSIST ? 17 : 0;
  auto ring_load = [&](unsigned s_) {
    return __hip_atomic_load(a.p_ring + (s_ & (unsigned)a.p_ring_mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  };
  // ---- synchronisation words, stage barrier and queues: kernel_a1_p4.hip's ----
  const unsigned flag_b = (unsigned)kFlagB;
  auto wait_word = [&](unsigned byte_addr, int want) {
    int tmp;
    asm volatile("1:\n\tds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_sub_u32 %0, %0, %2\n\tv_cmp_gt_i32 vcc, 0, %0\n\t"
                 "s_cbranch_vccz 2f\n\ts_sleep 1\n\ts_branch 1b\n2:"
                 : "=&v"(tmp)
                 : "v"(byte_addr), "v"(want)
                 : "vcc");
  };
  int bar_gen = 0;
  auto stage_barrier = [&]() {
    asm volatile("" ::: "memory");
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
  auto queue_put = [&](int q, int k, const f4& vx, const f4& vh, float vc, const i4& tok) {
    const unsigned slot = (unsigned)kQueueB + (unsigned)((q * 4 + w) * kp2::kSlotB);
    const unsigned prod = flag_b + (unsigned)(16 + 8 * q + w) * 4u, cons = flag_b + (unsigned)(16 + 8 * q + 4 + w) * 4u;
    wait_word(cons, k);
    asm volatile("" ::: "memory");
    lds_st4(lds, slot + lane16, vx);
    lds_st4(lds, slot + 1024u + lane16, vh);
    *reinterpret_cast<float*>(lds + slot + 2048u + (unsigned)lane * 4u) = vc;
    if (lane == 0)
      *reinterpret_cast<i4*>(lds + slot + 2304u) = tok;
    asm volatile("" ::: "memory");
    if (lane == 0)
      __hip_atomic_store(reinterpret_cast<int*>(lds + prod), k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
  };
  auto queue_take = [&](int q, int k, f4& vx, f4& vh, float& vc, i4& tok) {
    const unsigned slot = (unsigned)kQueueB + (unsigned)((q * 4 + w) * kp2::kSlotB);
    const unsigned prod = flag_b + (unsigned)(16 + 8 * q + w) * 4u, cons = flag_b + (unsigned)(16 + 8 * q + 4 + w) * 4u;
    wait_word(prod, k + 1);
    asm volatile("" ::: "memory");
    tok = *reinterpret_cast<const i4*>(lds + slot + 2304u);
    vx = lds_ld4(lds, slot + lane16);
    vh = lds_ld4(lds, slot + 1024u + lane16);
    vc = *reinterpret_cast<const float*>(lds + slot + 2048u + (unsigned)lane * 4u);
    asm volatile("" ::"v"(vx), "v"(vh), "v"(vc), "v"(tok) : "memory");
    if (lane == 0)
      __hip_atomic_store(reinterpret_cast<int*>(lds + cons), k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
  };
  // the two kh halves of a sum meet: lane i <-> lane i + 32 (the compiler's builtin for v_permlane32_swap loses the second
  // result in this ROCm, hence asm; the s_nop covers "matrix result -> vector read" and "vector write -> permlane read":
  // the hazard recogniser does not look into asm)
  auto combine = [&](f4& v) {
    f4 o = v;
    asm volatile("s_nop 7\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\tv_permlane32_swap_b32 %2, %6\n\t"
                 "v_permlane32_swap_b32 %3, %7\n\ts_nop 1"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
    v = v + o; // [lo | lo] + [hi | hi]
  };

  // the ring requests of the first kAhead jobs of every stage's first buffer (they depend on the state only)
  il::for_each_index(
    [&](auto s_tag) {
      constexpr int SS = decltype(s_tag)::value;
      if (S == SS)
      {
        constexpr int J0 = kp2::first_job(NST, SS);
        il::for_each_index(
          [&](auto u_tag) {
            constexpr int U = decltype(u_tag)::value;
            fetch(rows[U], std::integral_constant<int, J0 + U>{}, true, wp[U]);
          },
          std::make_integer_sequence<int, kp2::kAhead>{});
      }
    },
    std::make_integer_sequence<int, NST>{});
  // the tails of the first buffer: the last tail_rows(job) rows of every ring of this stage -> parity 1 (what buffer 0 reads)
  f4 tail0[MAXJ];
  il::for_each_index(
    [&](auto s_tag) {
      constexpr int SS = decltype(s_tag)::value;
      if (S == SS)
      {
        constexpr int J0 = kp2::first_job(NST, SS), NJS = kp2::first_job(NST, SS + 1) - J0;
        il::for_each_index(
          [&](auto u_tag) {
            constexpr int U = decltype(u_tag)::value;
            constexpr int JI = J0 + U, T = kp2::tail_rows(JI), RL = kp2::ring_len(JI), RO = kp2::ring_off(JI) * 4;
            const int ts = w * 64 + lane; // item = (row ts / 2, quad ts % 2)
            int row = wp[U] - T + (ts >> 1);
            row += row < 0 ? RL : 0;
            const int idx = (T > 0 && ts < 2 * T) ? row : kp2::kNoRow;
            tail0[U] = kp2_sb_load4(rs, idx, (ts & 1) * 16, RO, 0);
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
        unsigned long long v = ring_load(na);
        if ((unsigned)(v >> 32) != na + 1 && a.p_grace > 0)
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
    inp = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_in, frame * 4, uni((int)boff0), kInAux));
  // the weights and the first tails land in LDS
#pragma unroll
  for (int i = 0; i < kT4; i++)
    if (i * NT + tid < kW4)
      lds_st4(lds, (unsigned)kp2::kTilesB + (unsigned)(i * NT + tid) * 16u, tl4[i]);
  il::for_each_index(
    [&](auto s_tag) {
      constexpr int SS = decltype(s_tag)::value;
      if (S == SS)
      {
        constexpr int J0 = kp2::first_job(NST, SS), NJS = kp2::first_job(NST, SS + 1) - J0;
        il::for_each_index(
          [&](auto u_tag) {
            constexpr int U = decltype(u_tag)::value;
            constexpr int JI = J0 + U, T = kp2::tail_rows(JI);
            constexpr unsigned TB1 = (unsigned)kp2::tail_b(NST, JI, 1);
            const int ts = w * 64 + lane;
            if (T > 0 && ts < 2 * T)
              lds_st4(lds, TB1 + (unsigned)(ts >> 1) * (unsigned)kp2::kRowB + (unsigned)(ts & 1) * 16u, tail0[U]);
          },
          std::make_integer_sequence<int, NJS>{});
      }
    },
    std::make_integer_sequence<int, NST>{});
  lds_barrier();

  f4 x = {0.f, 0.f, 0.f, 0.f}, head = {0.f, 0.f, 0.f, 0.f};
  int nvalid = kBlock;
  float cond = 0.0f;
  unsigned long long spec_cmd = 0;
  float inp_spec = 0.0f;
  bool more = false;
  unsigned boff = 0;
  int par = 0; // parity of the buffer this wave's stage is working on (its jobs read the tails of parity par ^ 1, publish into par)
  // per-lane addresses: this lane's row of the stage's published buffers — to read its operand slice (minus one buffer of
  // rows: a tap's lookback L goes in as the immediate offset (64 - L) rows) and to publish its quad —, its row of the
  // wave's scratch, and frame * pitch (+ slice / quad) for the tails
  unsigned rd_b[3], wr_b[3];
#pragma unroll
  for (int b_ = 0; b_ < 3; b_++)
  {
    const unsigned base = (unsigned)kp2::kPubB + (unsigned)((S * 3 + b_) * kp2::kBufB);
    rd_b[b_] = base + (unsigned)frame * (unsigned)kp2::kRowB + opnd_b - (unsigned)(kBlock * kp2::kRowB);
    wr_b[b_] = base + (unsigned)frame * (unsigned)kp2::kRowB + quad_b;
  }
  constexpr unsigned kScratchB = (unsigned)kp2::scratch_b(NST);
  const unsigned scr_base = kScratchB + (unsigned)((S * 4 + w) * 16 * kp2::kRowB) + (unsigned)nn * (unsigned)kp2::kRowB;
  const unsigned scr_w = scr_base + quad_b, scr_r = scr_base + opnd_b;
  const unsigned f_opnd = (unsigned)frame * (unsigned)kp2::kRowB + opnd_b, f_quad = (unsigned)frame * (unsigned)kp2::kRowB + quad_b;

  // the value a job (or the stage) publishes for job NEXT: into published buffer WB and, for the lanes of the last
  // tail_rows(NEXT) frames, into NEXT's tail of this buffer's parity
  auto publish = [&](const f4& v, auto next_tag, auto wb_tag) {
    constexpr int NEXT = decltype(next_tag)::value, WB = decltype(wb_tag)::value;
    constexpr int T = kp2::tail_rows(NEXT);
    if (pub_lane)
      lds_st4(lds, wr_b[WB], v);
    if constexpr (T > 0)
    {
      constexpr unsigned TB0 = (unsigned)kp2::tail_b(NST, NEXT, 0) - (unsigned)((kBlock - T) * kp2::kRowB),
                         TB1 = (unsigned)kp2::tail_b(NST, NEXT, 1) - (unsigned)((kBlock - T) * kp2::kRowB);
      const unsigned tb = par ? TB1 : TB0;
      if (pub_lane && frame >= kBlock - T)
        lds_st4(lds, tb + f_quad, v);
    }
  };

  auto job = [&](auto j_tag, auto s_tag) {
    constexpr int JI = decltype(j_tag)::value;
    constexpr int SS = decltype(s_tag)::value;
    constexpr int J0 = kp2::first_job(NST, SS), J1 = kp2::first_job(NST, SS + 1), NJS = J1 - J0;
    constexpr int U = JI - J0;
    constexpr bool HEAD = JI == kp2::kLayers;
    constexpr int K = kp2::kKs[JI], D = kp2::kDs[JI], RL = kp2::ring_len(JI), T = kp2::tail_rows(JI);
    constexpr int RO = kp2::ring_off(JI) * 4, TAP0 = kp2::tap0(JI);
    constexpr int RB = U == 0 ? 0 : 1 + ((U - 1) & 1); // the published buffer this job reads (0 = E)
    constexpr int WB = 1 + (U & 1); // ... and the one it publishes into
    __builtin_amdgcn_sched_barrier(0);
    // (a) the job's input -> its history ring
    const int wpj = wp[U];
    {
      const unsigned v = (unsigned)(wpj + app_idx);
      const int widx = (int)min(v, v - (unsigned)RL);
      kp2_sb_store4(HEAD ? head : x, rs, widx, (int)quad_b, RO, WT && !PERSIST ? 17 : 0);
    }
    // (b) constants (zero in the kh = 1 lanes: the sum's seed enters once)
    const f4 bv = lds_ld4(lds, (unsigned)(kp2::kConstB + JI * 192) + (unsigned)g * 16u);
    f4 acc;
    if constexpr (HEAD)
      acc = bv;
    else
    {
      const f4 mv = lds_ld4(lds, (unsigned)(kp2::kConstB + JI * 192 + 64) + (unsigned)g * 16u);
      acc = __builtin_elementwise_fma(mv, f4{cond, cond, cond, cond}, bv);
    }
    if constexpr (PERSIST && JI == 1)
      spec_cmd = ring_load(na + 1);
    if constexpr (PERSIST && JI == 3)
    {
      const bool hit = (unsigned)(spec_cmd >> 32) == na + 2;
      const int soff = uni(hit ? (int)((unsigned)spec_cmd * 4u) : 0);
      inp_spec = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_in, hit ? frame * 4 : (int)kOob, soff, kInAux));
    }
    // (c) the taps, oldest first
    {
      Rows& Rj = rows[U];
      // this lane's row in the job's tail of the previous buffer's parity, minus one buffer of rows like rd_b
      unsigned tl_b = 0;
      if constexpr (T > 0)
      {
        constexpr unsigned TB0 = (unsigned)kp2::tail_b(NST, JI, 0) + (unsigned)((T - kBlock) * kp2::kRowB),
                           TB1 = (unsigned)kp2::tail_b(NST, JI, 1) + (unsigned)((T - kBlock) * kp2::kRowB);
        tl_b = (par ? TB0 : TB1) + f_opnd;
      }
      int hj = 0;
#pragma unroll
      for (int j = 0; j < K; j++)
      {
        const int L = (K - 1 - j) * D;
        const f4 tt = lds_ld4(lds, (unsigned)(kp2::kTilesB + (TAP0 + j) * 256) + cls16);
        f4 bq;
        if (L >= kBlock)
          bq = Rj.r[hj++];
        else if (L == 0)
          bq = *reinterpret_cast<const f4*>(lds + rd_b[RB] + (unsigned)(kBlock * kp2::kRowB));
        else
        {
          const unsigned ra = frame >= L ? rd_b[RB] : tl_b; // inside the buffer: the published rows; before it: the tail
          bq = *reinterpret_cast<const f4*>(lds + ra + (unsigned)((kBlock - L) * kp2::kRowB));
        }
#pragma unroll
        for (int c = 0; c < 4; c++)
          acc = __builtin_amdgcn_mfma_f32_4x4x1f32(tt[c], bq[c], acc, 0, 0, 0);
      }
    }
    // (d) the ring requests of the job kAhead jobs on (kernel_kp.hip)
    {
      constexpr int TU = (U + kp2::kAhead) % NJS;
      constexpr bool NEXT = U + kp2::kAhead >= NJS;
      fetch(rows[TU], std::integral_constant<int, J0 + TU>{}, NEXT ? more : true, wp[TU]);
    }
    combine(acc);
    // (e) epilogue
    if constexpr (HEAD)
    {
      const float yout = head_scale * acc[0];
      const bool ok = g == 0 && frame < nvalid;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yout), rsrc_out, ok ? frame * 4 : (int)kOob, uni((int)boff),
                                            PERSIST && !kOutHost ? 17 : 0);
    }
    else
    {
      f4 z;
      if constexpr (ACT_T == kp2::kActLeakyMax)
        z = __builtin_elementwise_max(acc, acc * act_p0);
      else
        z = act4<ACT_T>(act, acc, act_p0);
      head += z;
      asm volatile("" : "+v"(head));
      // the 1x1 needs channels 4 kh .. 4 kh + 3 of z for the lane's frame: through the wave's scratch rows (one wave's LDS
      // operations execute in order)
      if (pub_lane)
        lds_st4(lds, scr_w, z);
      const f4 b1v = lds_ld4(lds, (unsigned)(kp2::kConstB + JI * 192 + 128) + (unsigned)g * 16u);
      const f4 t1 = lds_ld4(lds, (unsigned)(kp2::kW1B + JI * 256) + cls16);
      const f4 zq = lds_ld4(lds, scr_r);
      f4 y = x + b1v; // (b1v is zero in the kh = 1 lanes; x must not enter twice)
      y = pub_lane ? y : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 4; c++)
        y = __builtin_amdgcn_mfma_f32_4x4x1f32(t1[c], zq[c], y, 0, 0, 0);
      combine(y);
      x = y;
      if constexpr (U + 1 < NJS)
      {
        if constexpr (JI + 1 == kp2::kLayers)
          publish(head, std::integral_constant<int, JI + 1>{}, std::integral_constant<int, WB>{});
        else
          publish(x, std::integral_constant<int, JI + 1>{}, std::integral_constant<int, WB>{});
        stage_barrier();
      }
    }
    {
      int np = wpj + nvalid;
      np -= np >= RL ? RL : 0;
      wp[U] = np;
    }
  };

  // ---- the stage loops (kernel_a1_p4.hip: run) ----
  auto run = [&](auto s_tag) {
    constexpr int SS = decltype(s_tag)::value;
    constexpr int J0 = kp2::first_job(NST, SS), NJS = kp2::first_job(NST, SS + 1) - J0;
    static_assert(NJS > kp2::kAhead, "a stage's ring requests run kAhead jobs ahead inside the stage");
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
        queue_take(QIN, k, x, head, cond, tok);
        boff = (unsigned)uni(tok[0]);
        nvalid = uni(tok[1]);
        exit_tok = uni(tok[2]) != 0;
        more = PERSIST || uni(tok[3]) != 0;
      }
      auto hand_over = [&](bool is_exit) {
        queue_put(QOUT < 0 ? 0 : QOUT, k, x, head, cond, i4{(int)boff, nvalid, is_exit ? 1 : 0, more ? 1 : 0});
      };
      if (exit_tok)
      {
        if constexpr (!LAST)
          hand_over(true);
        break;
      }
      par = k & 1;
      if constexpr (!PERSIST)
      {
        if (nvalid != kBlock) // a ragged last block: only its frames are appended (nothing reads its tails: it is the last)
          app_idx = app_of(nvalid);
      }
      if constexpr (FIRST)
      {
        cond = inp;
        if constexpr (!PERSIST)
          inp = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_in, frame * 4, uni((k + 1) * (kBlock * 4)), 0));
        x = rech * cond;
        head = f4{0.f, 0.f, 0.f, 0.f};
      }
      // the stage's input -> its buffer E (and the tail of its first job)
      publish(x, std::integral_constant<int, J0>{}, std::integral_constant<int, 0>{});
      stage_barrier();
      il::for_each_index([&](auto u_tag) { job(std::integral_constant<int, J0 + decltype(u_tag)::value>{}, s_tag); },
                         std::make_integer_sequence<int, NJS>{});
      if constexpr (!LAST)
        hand_over(false);
      else if constexpr (PERSIST)
      {
        done++;
        if (w == 0 && lane == 0 && (done & 15u) == 0u)
          __hip_atomic_store(a.p_prog + blockIdx.x, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      if constexpr (FIRST)
      {
        if constexpr (PERSIST)
        {
          if (w == 0)
          {
            const unsigned tag = na + 2u;
            unsigned long long v = spec_cmd;
            if ((unsigned)(v >> 32) != tag)
            {
              v = ring_load(tag - 1u);
              const long long t_end = (long long)wall_clock64() + 300; // 3 us of the 100 MHz clock
              while ((unsigned)(v >> 32) != tag && (long long)wall_clock64() < t_end)
              {
                __builtin_amdgcn_s_sleep(16);
                v = ring_load(tag - 1u);
              }
            }
            if (lane == 0)
            {
              flags[48 + 2 * (k & 1)] = (int)(unsigned)v;
              flags[48 + 2 * (k & 1) + 1] = (unsigned)(v >> 32) == tag ? 1 : 0;
            }
          }
          stage_barrier();
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

  il::for_each_index(
    [&](auto s_tag) {
      constexpr int SS = decltype(s_tag)::value;
      if (S == SS && w == 0)
      {
        constexpr int J0 = kp2::first_job(NST, SS), NJS = kp2::first_job(NST, SS + 1) - J0;
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
constexpr int kKp2Stages = 3;

template <int ACT_T, bool WT, bool PERSIST = false>
hipError_t launch_kp2_inst(const A1Args& a, int n_blocks, hipStream_t stream)
{
  static DynamicLdsLimit lds_limit;
  constexpr int lds_bytes = kp2::lds_bytes(kKp2Stages);
  const hipError_t e = lds_limit.ensure(reinterpret_cast<const void*>(&nam_kp2_kernel<ACT_T, WT, PERSIST, kKp2Stages>), lds_bytes);
  if (e != hipSuccess)
    return e;
  hipLaunchKernelGGL((nam_kp2_kernel<ACT_T, WT, PERSIST, kKp2Stages>), dim3(n_blocks), dim3(kKp2Stages * 256), lds_bytes, stream, a.blob, a);
  return hipGetLastError();
}
template <int ACT_T>
hipError_t launch_kp2_act(const A1Args& a, int n_blocks, hipStream_t stream)
{
  if (a.p_ring)
    return a.p_out_host != 0 ? launch_kp2_inst<ACT_T, true, true>(a, n_blocks, stream) : launch_kp2_inst<ACT_T, false, true>(a, n_blocks, stream);
  const bool wt = a.n_frames <= 2 * kBlock;
  return wt ? launch_kp2_inst<ACT_T, true>(a, n_blocks, stream) : launch_kp2_inst<ACT_T, false>(a, n_blocks, stream);
}
} // namespace

// a.tiles_off: blob offset (floats) of the kernel's weight block [tap tiles | 1x1 tiles | constants] (plan.cpp: build_a1_kp);
// a.r1_off: of the rechannel column per lane group [4][4]; a.act: the array's activation
hipError_t launch_kp2(const A1Args& a, int n_blocks, int act, hipStream_t stream)
{
  if (act == ACT_LEAKYRELU && a.act_p0 <= 1.0f) // (A2: 0.01)
    return launch_kp2_act<kp2::kActLeakyMax>(a, n_blocks, stream);
  return launch_kp2_act<-1>(a, n_blocks, stream);
}

} // namespace namhip
