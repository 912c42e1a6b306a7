// kernel_kp.hip — nam_kp_kernel: the A2 topology (kp_table.h) as a PIPELINE OF WAVE SETS, everything about a layer known at
// compile time.
#include "device_common.h"
#include "il_common.h"
#include "kp_table.h"

namespace namhip
{

// ================================================================================================
// nam_kt_mfma_kernel (kernel_kt_mfma.hip; read its header first) runs A2-Full — 8 channels, 23 layers of 6 or 15 taps at
// dilations up to 239, a 16-tap head rechannel: the shape of the reference's own fused path, NAM/wavenet/a2_fast.cpp —
// with one wavefront per SIMD from a run-time chunk table: ~340 instructions around every 12 matrix instructions and
// every latency exposed (35 us per 64-frame buffer at 256 streams, 9.7 k xRT). This kernel is what nam_a1_p4_kernel
// (kernel_a1_p4.hip) is to nam_a1_p2_kernel, for that topology:
//   * the 24 jobs of a buffer (23 layers + the head rechannel) are cut into NST = 3 stages of consecutive jobs, balanced
//     on matrix instructions (kp::first_job: 0-8 | 9-15 | 16-23 for A2); stage s is a set of four waves working on buffer
//     k - s while stage 0 works on buffer k: twelve waves per stream, three per SIMD. Wave w of every stage owns frames
//     [16 w, 16 w + 16) in nam_kt_mfma_kernel's lane layout (half layout: lane (g, n) holds four output channels of frame
//     16 w + n, feeds channels in_chan(g, 0..1) as its B operand); the state (rings, write positions), the tap tiles and
//     the constants are that kernel's, so the two alternate freely between launches of one stream;
//   * stage to stage: the one-slot LDS queues of nam_a1_p4_kernel (x, head accumulator, input sample, token), wave w to
//     wave w; inside a stage every layer ends with "publish the layer output in LDS, stage barrier" (a tap reaches up to
//     63 frames back inside the buffer: any wave's rows) — four generation words polled from asm, never the hardware
//     barrier (it would stop all twelve waves);
//   * a tap's B operand is the lane's slice of frame t - L: from the published rows when that frame lies inside the
//     buffer, from the layer's history ring in HBM otherwise — requested TWO JOBS AHEAD by row index through a strided
//     buffer descriptor (lanes whose frame is inside the buffer carry an index no descriptor holds: no traffic, the
//     load returns 0) and added to the LDS operand (row 0 of a published buffer is zero: lanes whose frame lies before
//     the buffer read it). Which of the two a tap needs is known at compile time: lookbacks >= 64 have no LDS read,
//     lookback 0 no request;
//   * tap tiles (92 KB) and the 1x1 tiles / constants (16 KB) live in LDS for the whole launch.
// Per layer of 6 taps and wave: 14 matrix instructions, ~25 LDS and 6 vector-memory instructions, ~90 others.
// Sums: one accumulator chain per layer seeded with bias + mixin * input (taps oldest first, then the current frame),
// one chain for the 1x1 seeded with x + bias — model.cpp:183-393's order up to the association inside a dot product.
// ================================================================================================
using kp_i4 = __attribute__((ext_vector_type(4))) int;
__device__ mf::f2 kp_sb_load2(kp_i4 rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.v2f32");
__device__ void kp_sb_store4(mf::f4 v, kp_i4 rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.store.v4f32");

namespace kp
{
constexpr int kNoRow = 1 << 26; // a ring row index no descriptor holds: the access is dropped / returns 0
constexpr int kRows = 1 << 20; // num_records of the ring descriptor (rows); every real row index is far below
constexpr int kActLeakyMax = 100; // ACT_T of the LeakyReLU instantiation (slope <= 1)
#ifndef NAM_KP_AHEAD
#define NAM_KP_AHEAD 2
#endif
constexpr int kAhead = NAM_KP_AHEAD; // a job's ring rows are requested this many jobs earlier (of the same stage, wrapping to the next buffer)
constexpr int kRowB = (kC + 4) * 4; // LDS pitch of a published frame row
constexpr int kBufB = (kBlock + 1) * kRowB; // one published buffer: row 0 = zeros, row 1 + t = frame t
// LDS layout (bytes)
constexpr int kTilesB = 0; // tap tiles [chunk][3 pairs][64 lanes][4]
constexpr int kAuxB = kTilesB + kChunks * kTapsPerChunk * 64 * 2 * 4; // 1x1 tiles [layer][64 lanes][2] | constants [job][3][16]
constexpr int kAuxFloats = kLayers * 128 + kJobs * 48;
constexpr int kW1B(int l) { return kAuxB + l * 512; }
constexpr int kConstB(int job) { return kAuxB + kLayers * 512 + job * 192; }
constexpr int kPubB = kAuxB + kAuxFloats * 4; // per stage three published buffers: E (the stage's input), 0, 1 (alternating)
constexpr int pub_b(int stage, int buf) { return kPubB + (stage * 3 + buf) * kBufB; }
constexpr int flag_b(int nst) { return kPubB + nst * 3 * kBufB; } // 256 bytes of single-writer words (kernel_a1_p4.hip)
constexpr int queue_b(int nst) { return flag_b(nst) + 256; }
constexpr int kSlotB = 64 * 16 + 64 * 16 + 64 * 4 + 16;
constexpr int lds_bytes(int nst) { return queue_b(nst) + (nst - 1) * 4 * kSlotB; }
static_assert(lds_bytes(3) <= 160 * 1024, "kp LDS layout");
static_assert(kBufB % 16 == 0 && kAuxB % 16 == 0, "kp LDS alignment");
// tap j of a job: byte offset of its tile pair record inside the tile area, and which half of the record it is
constexpr int tile_b(int job, int j) { return ((chunk0(job) + j / kTapsPerChunk) * 3 + (j % kTapsPerChunk) / 2) * 1024; }
} // namespace kp

template <int ACT_T, bool WT, bool PERSIST, int NST>
__global__ __launch_bounds__(NST * 256) void nam_kp_kernel(const float* __restrict__ blob, const A1Args a)
{
  using namespace mf;
  using il::kOob;
  using i4 = kp_i4;
  constexpr int NJ = kp::kJobs, MAXJ = kp::max_jobs(NST), MAXK = kp::max_k();
  extern __shared__ __attribute__((aligned(16))) float lds_kp[];
  char* const lds = reinterpret_cast<char*>(lds_kp);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wall = uni(tid >> 6);
  const int S = wall >> 2; // stage
  const int w = wall & 3;
  const int stream = a.stream_map ? a.stream_map[blockIdx.x] : (int)blockIdx.x;
  float* st = a.state + (size_t)stream * a.state_stride;
  const int n_blocks = PERSIST ? (1 << 30) : (a.n_frames + kBlock - 1) / kBlock;

  const int g = lane >> 4;
  const int frame = 16 * w + (lane & 15); // this lane's frame inside the buffer
  const float* in = a.in ? a.in + (size_t)stream * a.io_stride : nullptr;
  float* out = a.out ? a.out + (size_t)stream * a.io_stride : nullptr;
  const float head_scale = a.head_scale;
  const float act_p0 = a.act_p0;
  const int act = a.act; // (only read by the run-time-dispatch instantiation)
  const unsigned quad_b = (unsigned)g * 16u; // the quad this lane publishes / appends (lanes g < 2: channels 4 g .. 4 g + 3)
  const unsigned opnd_b = (unsigned)((g & 1) * 16 + (g >> 1) * 8); // its B-operand slice of a frame row
  const bool pub_lane = g < 2;
  const unsigned lane16 = (unsigned)lane * 16u;
  const int io_bytes = PERSIST ? 0x7ffffff0 : a.n_frames * 4;
  const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc((void*)(in ? in : st), 0, in ? io_bytes : 0, 0x00020000);
  const auto rsrc_out = __builtin_amdgcn_make_buffer_rsrc((void*)(out ? out : st), 0, out ? io_bytes : 0, 0x00020000);
  const unsigned long long in_addr = (unsigned long long)(in ? in : st);
  const i4 in_desc = {uni((int)(unsigned)in_addr), uni((int)(unsigned)(in_addr >> 32) & 0xffff), in ? io_bytes : 0, 0x00020000};
  int* wpos_tbl = reinterpret_cast<int*>(st);
  const int wposv = lane < NJ ? wpos_tbl[lane] : 0; // lane r = write position of ring r (ring r = job r) at launch
  constexpr int kFlagB = kp::flag_b(NST), kQueueB = kp::queue_b(NST);
  int* const flags = reinterpret_cast<int*>(lds + kFlagB);

  // ---- tap tiles, 1x1 tiles and constants -> LDS, once per launch, by every wave ----
  constexpr int NT = NST * 256;
  constexpr int kTile4 = kp::kChunks * kp::kTapsPerChunk * 64 * 2 / 4, kAux4 = kp::kAuxFloats / 4; // 16-byte records
  constexpr int kT4 = (kTile4 + NT - 1) / NT, kA4 = (kAux4 + NT - 1) / NT;
  f4 tl4[kT4], ax4[kA4];
  {
    const f4* __restrict__ tsrc = reinterpret_cast<const f4*>(blob + a.tiles_off);
    const f4* __restrict__ asrc = reinterpret_cast<const f4*>(blob + a.consts_off);
#pragma unroll
    for (int i = 0; i < kT4; i++)
      tl4[i] = tsrc[min(i * NT + tid, kTile4 - 1)];
#pragma unroll
    for (int i = 0; i < kA4; i++)
      ax4[i] = asrc[min(i * NT + tid, kAux4 - 1)];
  }
  const f4 rech = *reinterpret_cast<const f4*>(blob + a.r1_off + g * 4);

  // The stream's rings through ONE descriptor with the row pitch (32 bytes) as the stride: an access names its row by
  // index and its ring by the scalar offset; kNoRow drops it.
  const unsigned long long st_addr = (unsigned long long)st;
  const i4 rs = {uni((int)(unsigned)st_addr), uni((int)((unsigned)(st_addr >> 32) & 0xffffu) | ((kp::kC * 4) << 16)), kp::kRows, 0x00020000};
  auto app_of = [&](int nv) { return (pub_lane && frame < nv) ? frame : kp::kNoRow; };
  int app_idx = app_of(kBlock); // this lane's frame as a row offset when it appends (kNoRow: it holds no channels to append)
  // the write positions of this stage's rings as SCALARS (wp[u] = ring of job J0 + u)
  int wp[MAXJ];
  il::for_each_index(
    [&](auto s_tag) {
      constexpr int SS = decltype(s_tag)::value;
      if (S == SS)
      {
        constexpr int J0 = kp::first_job(NST, SS), NJS = kp::first_job(NST, SS + 1) - J0;
#pragma unroll
        for (int u = 0; u < MAXJ; u++)
          wp[u] = u < NJS ? __builtin_amdgcn_readlane(wposv, J0 + (u < NJS ? u : 0)) : 0;
      }
    },
    std::make_integer_sequence<int, NST>{});
  // the history operands of job TJ (its taps with a lookback) for the buffer that starts at write position `wpj` of its
  // ring; `valid`: wave-uniform
  struct Rows
  {
    f2 r[MAXK];
  };
  auto fetch = [&](Rows& R_, auto tj_tag, bool valid, int wpj) {
    constexpr int TJ = decltype(tj_tag)::value;
    constexpr int K = kp::kKs[TJ], D = kp::kDs[TJ], RL = kp::ring_len(TJ);
#pragma unroll
    for (int j = 0; j < K - 1; j++)
    {
      const int L = (K - 1 - j) * D; // > 0
      // row of lane frame f: (wpj - L + f) mod RL — the lane-independent part on the scalar unit
      int sb_ = wpj - L;
      sb_ += sb_ < 0 ? RL : 0;
      sb_ = valid ? sb_ : kp::kNoRow;
      const int fq = L >= kBlock ? frame : (frame < L ? frame : kp::kNoRow); // only frames that lie before the buffer
      const unsigned v = (unsigned)(sb_ + fq);
      const int idx = (int)min(v, v - (unsigned)RL);
      R_.r[j] = kp_sb_load2(rs, idx, (int)opnd_b, kp::ring_off(TJ) * 4, 0);
    }
  };
  Rows rows[MAXJ];
  float inp = 0.0f;

  constexpr bool kOutHost = PERSIST && WT; // (kernel_a1_p4.hip: a session whose results go to host memory)
  constexpr int kInAux = PERSIST ? 17 : 0; // session inputs bypass the caches (the caller may rewrite the buffer between commands)
  auto ring_load = [&](unsigned s_) {
    return __hip_atomic_load(a.p_ring + (s_ & (unsigned)a.p_ring_mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  };
  // ---- synchronisation words, stage barrier and queues: kernel_a1_p4.hip's (see there for why each is what it is) ----
  const unsigned flag_b = (unsigned)kFlagB;
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
    asm volatile("" ::: "memory"); // (the published rows' plain stores / loads stay on their side)
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
    const unsigned slot = (unsigned)kQueueB + (unsigned)((q * 4 + w) * kp::kSlotB);
    const unsigned prod = flag_b + (unsigned)(16 + 8 * q + w) * 4u, cons = flag_b + (unsigned)(16 + 8 * q + 4 + w) * 4u;
    wait_word(cons, k); // the slot is free once buffer k - 1 has been taken out of it
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
    const unsigned slot = (unsigned)kQueueB + (unsigned)((q * 4 + w) * kp::kSlotB);
    const unsigned prod = flag_b + (unsigned)(16 + 8 * q + w) * 4u, cons = flag_b + (unsigned)(16 + 8 * q + 4 + w) * 4u;
    wait_word(prod, k + 1);
    asm volatile("" ::: "memory");
    tok = *reinterpret_cast<const i4*>(lds + slot + 2304u);
    vx = lds_ld4(lds, slot + lane16);
    vh = lds_ld4(lds, slot + 1024u + lane16);
    vc = *reinterpret_cast<const float*>(lds + slot + 2048u + (unsigned)lane * 4u);
    asm volatile("" ::"v"(vx), "v"(vh), "v"(vc), "v"(tok) : "memory"); // (the values are in registers: the slot may be reused)
    if (lane == 0)
      __hip_atomic_store(reinterpret_cast<int*>(lds + cons), k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
  };

  // the ring requests of the first kAhead jobs of every stage's first buffer (they depend on the state only)
  il::for_each_index(
    [&](auto s_tag) {
      constexpr int SS = decltype(s_tag)::value;
      if (S == SS)
      {
        constexpr int J0 = kp::first_job(NST, SS);
        il::for_each_index(
          [&](auto u_tag) {
            constexpr int U = decltype(u_tag)::value;
            fetch(rows[U], std::integral_constant<int, J0 + U>{}, true, wp[U]);
          },
          std::make_integer_sequence<int, kp::kAhead>{});
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
    inp = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_in, frame * 4, uni((int)boff0), kInAux));
  // the weights (requested before the ring rows, so they are here first); row 0 of every published buffer is zero
#pragma unroll
  for (int i = 0; i < kT4; i++)
    if (i * NT + tid < kTile4)
      lds_st4(lds, (unsigned)kp::kTilesB + (unsigned)(i * NT + tid) * 16u, tl4[i]);
#pragma unroll
  for (int i = 0; i < kA4; i++)
    if (i * NT + tid < kAux4)
      lds_st4(lds, (unsigned)kp::kAuxB + (unsigned)(i * NT + tid) * 16u, ax4[i]);
  if (tid < NST * 3 * (kp::kRowB / 4))
    *reinterpret_cast<float*>(lds + kp::kPubB + (tid / (kp::kRowB / 4)) * kp::kBufB + (tid % (kp::kRowB / 4)) * 4) = 0.0f;
  lds_barrier();

  f4 x = {0.f, 0.f, 0.f, 0.f}, head = {0.f, 0.f, 0.f, 0.f};
  int nvalid = kBlock; // frames of the buffer this wave's stage is working on
  float cond = 0.0f;
  unsigned long long spec_cmd = 0; // PERSIST, stage 0: this wave's early look at the next command ...
  float inp_spec = 0.0f; // ... and the input sample it requested on a hit
  bool more = false; // this wave's stage has another buffer behind the current one (its requests are made for it)
  unsigned boff = 0; // byte offset of the current buffer in the stream's row
  // this lane's row of a published buffer: byte address of (frame + 1, operand slice) and of row 0, per buffer of the stage
  unsigned rd_b[3], rd_lo[3], wr_b[3];
#pragma unroll
  for (int b_ = 0; b_ < 3; b_++)
  {
    const unsigned base = (unsigned)kp::kPubB + (unsigned)((S * 3 + b_) * kp::kBufB);
    rd_lo[b_] = base + opnd_b;
    rd_b[b_] = base + (unsigned)(frame + 1) * (unsigned)kp::kRowB + opnd_b;
    wr_b[b_] = base + (unsigned)(frame + 1) * (unsigned)kp::kRowB + quad_b;
  }

  // One job = one layer (or the head rechannel), everything about it known at compile time. model.cpp:183-393:
  // z = act(conv(x) + mixin(cond)); head += z; x += layer1x1(z); model.cpp:513-531 for the head rechannel.
  auto job = [&](auto j_tag, auto s_tag) {
    constexpr int JI = decltype(j_tag)::value;
    constexpr int SS = decltype(s_tag)::value;
    constexpr int J0 = kp::first_job(NST, SS), J1 = kp::first_job(NST, SS + 1), NJS = J1 - J0;
    constexpr int U = JI - J0; // position in the stage
    constexpr bool HEAD = JI == kp::kLayers;
    constexpr int K = kp::kKs[JI], D = kp::kDs[JI], RL = kp::ring_len(JI);
    constexpr int RB = U == 0 ? 0 : 1 + ((U - 1) & 1); // the published buffer this job reads (0 = E)
    constexpr int WB = 1 + (U & 1); // ... and the one it publishes into
    __builtin_amdgcn_sched_barrier(0);
    // (a) the job's input (what was published for it) -> its history ring: row (position + frame) mod R
    const int wpj = wp[U];
    {
      const unsigned v = (unsigned)(wpj + app_idx);
      const int widx = (int)min(v, v - (unsigned)RL);
      kp_sb_store4(HEAD ? head : x, rs, widx, (int)quad_b, kp::ring_off(JI) * 4, WT && !PERSIST ? 17 : 0);
    }
    // (b) constants
    const f4 bv = lds_ld4(lds, (unsigned)kp::kConstB(JI) + quad_b);
    f4 acc;
    if constexpr (HEAD)
      acc = bv;
    else
    {
      const f4 mv = lds_ld4(lds, (unsigned)kp::kConstB(JI) + 64u + quad_b);
      acc = __builtin_elementwise_fma(mv, f4{cond, cond, cond, cond}, bv);
    }
    // persistent session, stage 0: every wave looks at the next ring slot in job 1 and, when the command is already
    // there, requests the next buffer's input sample from it in job 3 (unconditional load, out-of-range offset on a miss)
    if constexpr (PERSIST && JI == 1)
      spec_cmd = ring_load(na + 1);
    if constexpr (PERSIST && JI == 3)
    {
      const bool hit = (unsigned)(spec_cmd >> 32) == na + 2;
      const int soff = uni(hit ? (int)((unsigned)spec_cmd * 4u) : 0);
      inp_spec = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_in, hit ? frame * 4 : (int)kOob, soff, kInAux));
    }
    // (c) the taps, oldest first: ONE chain
    {
      Rows& Rj = rows[U];
#pragma unroll
      for (int p = 0; p < (K + 1) / 2; p++)
      {
        const f4 tt = lds_ld4(lds, (unsigned)(kp::kTilesB + kp::tile_b(JI, 2 * p)) + lane16);
#pragma unroll
        for (int h = 0; h < 2; h++)
        {
          const int j = 2 * p + h;
          if (j < K)
          {
            const int L = (K - 1 - j) * D;
            f2 bq;
            if (L == 0)
              bq = *reinterpret_cast<const f2*>(lds + rd_b[RB]);
            else if (L >= kBlock)
              bq = Rj.r[j];
            else
            {
              const unsigned ra = (unsigned)max((int)rd_b[RB] - L * kp::kRowB, (int)rd_lo[RB]);
              // exactly one of the two is the operand, the other is 0
              bq = *reinterpret_cast<const f2*>(lds + ra) + Rj.r[j];
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(tt[2 * h], bq[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(tt[2 * h + 1], bq[1], acc, 0, 0, 0);
          }
        }
      }
    }
    // (d) the ring requests of the job kAhead jobs on (of this stage; behind its last job: the first jobs of the stage's
    // NEXT buffer — those jobs have run for this buffer, so their write positions already stand at the next one, and the
    // barriers since then order this request behind every wave's appends). Issued behind this job's tap products: the
    // rows requested here replace operands that are dead by now.
    {
      constexpr int TU = (U + kp::kAhead) % NJS;
      constexpr bool NEXT = U + kp::kAhead >= NJS;
      fetch(rows[TU], std::integral_constant<int, J0 + TU>{}, NEXT ? more : true, wp[TU]);
    }
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
      const f4 b1v = lds_ld4(lds, (unsigned)kp::kConstB(JI) + 128u + quad_b);
      const f2 w1 = *reinterpret_cast<const f2*>(lds + (unsigned)kp::kW1B(JI) + (unsigned)lane * 8u);
      f4 z;
      if constexpr (ACT_T == kp::kActLeakyMax) // LeakyReLU with a slope <= 1: max(v, slope * v) (launch_kp checks the slope)
        z = __builtin_elementwise_max(acc, acc * act_p0);
      else
        z = act4<ACT_T>(act, acc, act_p0);
      head += z;
      asm volatile("" : "+v"(head)); // (pin the accumulator: kernel_a1_p2.hip)
      f4 y = x + b1v;
      y = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[0], z[0], y, 0, 0, 0);
      y = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[1], z[1], y, 0, 0, 0);
      x = y;
      if constexpr (U + 1 < NJS)
      {
        // the next job of this stage reads it from LDS (the last layer's successor, the head rechannel, reads the head
        // accumulator); the stage's last job hands x over through the queue instead
        if (pub_lane)
          lds_st4(lds, wr_b[WB], JI + 1 == kp::kLayers ? head : x);
        stage_barrier();
      }
    }
    // this ring moves on by the buffer's frames (scalar unit)
    {
      int np = wpj + nvalid;
      np -= np >= RL ? RL : 0;
      wp[U] = np;
    }
  };

  // ---- the stage loops (kernel_a1_p4.hip: run) ----
  auto run = [&](auto s_tag) {
    constexpr int SS = decltype(s_tag)::value;
    constexpr int J0 = kp::first_job(NST, SS), NJS = kp::first_job(NST, SS + 1) - J0;
    static_assert(NJS > kp::kAhead, "a stage's ring requests run kAhead jobs ahead inside the stage");
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
        x = rech * cond;
        head = f4{0.f, 0.f, 0.f, 0.f};
      }
      // the stage's input -> its buffer E, for the taps of its first job
      if (pub_lane)
        lds_st4(lds, wr_b[0], x);
      stage_barrier();
      il::for_each_index([&](auto u_tag) { job(std::integral_constant<int, J0 + decltype(u_tag)::value>{}, s_tag); },
                         std::make_integer_sequence<int, NJS>{});
      if constexpr (!LAST)
        hand_over(false);
      else if constexpr (PERSIST)
      {
        done++;
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
              // the early look missed: look again; while the later stages still work, for a few microseconds (bounded:
              // the launch never waits for a command)
              v = ring_load(tag - 1u);
              const long long t_end = (long long)wall_clock64() + 100; // 1 us of the 100 MHz clock
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
          stage_barrier(); // wave 0's decision is the stage's (two decision slots by parity)
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
              // (rare: this wave's look came too early) load now and wait right here, inside the asm
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

  // the write positions of this stage's rings go back into the state (lane r of the stage's wave 0 = ring r)
  il::for_each_index(
    [&](auto s_tag) {
      constexpr int SS = decltype(s_tag)::value;
      if (S == SS && w == 0)
      {
        constexpr int J0 = kp::first_job(NST, SS), NJS = kp::first_job(NST, SS + 1) - J0;
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
      if (S == NST - 1 && w == 0) // one wave per workgroup asks for the write-back (kernel_a1_p4.hip)
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
constexpr int kKpStages = 3;

template <int ACT_T, bool WT, bool PERSIST = false>
hipError_t launch_kp_inst(const A1Args& a, int n_blocks, hipStream_t stream)
{
  static DynamicLdsLimit lds_limit; // per instantiation, tracked per device (kernels.h)
  constexpr int lds_bytes = kp::lds_bytes(kKpStages);
  const hipError_t e = lds_limit.ensure(reinterpret_cast<const void*>(&nam_kp_kernel<ACT_T, WT, PERSIST, kKpStages>), lds_bytes);
  if (e != hipSuccess)
    return e;
  hipLaunchKernelGGL((nam_kp_kernel<ACT_T, WT, PERSIST, kKpStages>), dim3(n_blocks), dim3(kKpStages * 256), lds_bytes, stream, a.blob, a);
  return hipGetLastError();
}
template <int ACT_T>
hipError_t launch_kp_act(const A1Args& a, int n_blocks, hipStream_t stream)
{
  if (a.p_ring) // persistent session (kernel_a1_p4.hip: launch_p4_shape)
    return a.p_out_host != 0 ? launch_kp_inst<ACT_T, true, true>(a, n_blocks, stream) : launch_kp_inst<ACT_T, false, true>(a, n_blocks, stream);
  const bool wt = a.n_frames <= 2 * kBlock; // short launches write ring appends through
  return wt ? launch_kp_inst<ACT_T, true>(a, n_blocks, stream) : launch_kp_inst<ACT_T, false>(a, n_blocks, stream);
}
} // namespace

// a.tiles_off / a.consts_off / a.r1_off: blob offsets (floats) of the K-tap kernel's tap tiles, of its LDS block (1x1 tiles |
// constants) and of the rechannel column; a.act: the array's activation
hipError_t launch_kp(const A1Args& a, int n_blocks, int act, hipStream_t stream)
{
  if (act == ACT_LEAKYRELU && a.act_p0 <= 1.0f) // (A2: 0.01)
    return launch_kp_act<kp::kActLeakyMax>(a, n_blocks, stream);
  return launch_kp_act<-1>(a, n_blocks, stream);
}

} // namespace namhip
