// kernel_a1_mfma.hip — nam_a1_mfma_kernel, the headline kernel.
#include "device_common.h"

namespace namhip
{

// ================================================================================================
// nam_a1_mfma_kernel — wave-specialised fp32-MFMA kernel: 8 wavefronts per stream, one job per LAYER.
//   waves 0-3 (compute): per job one barrier, 3 LDS reads on the critical path (two shifted taps + the input
//                        sample), 16-20 MFMAs, activation, publish x. The job's weight tiles and constants are
//                        already in registers: they are read from LDS one job ahead, in the shadow of the MFMAs.
//                        No vector-memory instructions except the output store.
//   waves 4-7 (movers) : per job drop the successor's prefetched history (3 x 16 B per lane) and the weight
//                        tiles of the job after it (16 B per lane) into the LDS double buffers, append the
//                        job's input rows (LDS window) to its HBM ring, and issue the loads of the job
//                        D + 1 ahead (D = the plan's ws_prefetch). At block boundaries they also materialise
//                        x0 = rechannel * input and the input samples in LDS.
// Each SIMD hosts one compute and one mover wave, so the mover's address arithmetic / memory instructions
// fill the issue slots the compute wave leaves between dependent MFMA / VALU instructions.
// Rechannel and head-rechannel steps ride on the neighbouring layer jobs (plan.h: CDesc / VDesc).
// ================================================================================================
namespace ws
{
using mf::f4;
constexpr int SC = kMfSC;
struct HSlot
{
  f4 h[2]; // the job's two history sets (plan.h, VDesc)
  f4 tile; // 16 B of the weight tiles of the job AFTER the one the history belongs to
  float inp; // input sample of frame hfr of the block the job belongs to
};
struct Ops // one job's register-resident operands (compute waves)
{
  f4 t[4]; // A tiles: conv tap 0,1,2 | layer1x1
  f4 xt; // extra tile (rechannel / head rechannel) when the job has one
  f4 bv, mv, b1v, ev; // conv bias | input mixin | 1x1 bias | extra (rechannel column or head bias)
};
} // namespace ws

template <int ACT_T, bool WT, bool PROF, int D>
__global__ __launch_bounds__(512) void nam_a1_mfma_kernel(const A1Plan* __restrict__ P, const float* __restrict__ blob,
                                                        const A1Args a)
{
  using namespace mf;
  using ws::HSlot;
  using ws::Ops;
  constexpr int SC = ws::SC;
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  char* const lds = reinterpret_cast<char*>(lds_f);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = uni(tid >> 6);
  const int stream = a.stream_map ? a.stream_map[blockIdx.x] : (int)blockIdx.x;
  float* st = a.state + (size_t)stream * a.state_stride;
  char* stb = reinterpret_cast<char*>(st);
  const int NJ = a.n_mjobs;
  const int n_blocks = (a.n_frames + kBlock - 1) / kBlock;
  const int total = n_blocks * NJ;
  constexpr int kUnroll = (D % 2 == 0) ? D : 2 * D; // both roles run a multiple of this many jobs (= barriers)
  const int total_pad = (total + kUnroll - 1) / kUnroll * kUnroll;
  const unsigned lds_tiles_b = (unsigned)a.lds_tiles_b, lds_cond_b = (unsigned)a.lds_cond_b;

  // constants table and extra tiles -> LDS, by the compute waves only (visible at the prologue barrier): the movers'
  // prologue is two dependent memory round trips (write positions -> first history sets) and must not queue behind
  // a third
  if (w < 4)
  {
    const float* __restrict__ csrc = blob + a.consts_off;
    const float* __restrict__ xsrc = blob + a.xt_off;
    constexpr int NC = kWsJobMax * 64 / 256, NX = kWsXtMax * 256 / 256;
    float cv[NC], xv[NX];
#pragma unroll
    for (int i = 0; i < NC; i++)
      cv[i] = csrc[tid + 256 * i]; // the blob tables are padded to their maximum sizes
#pragma unroll
    for (int i = 0; i < NX; i++)
      xv[i] = xsrc[tid + 256 * i];
#pragma unroll
    for (int i = 0; i < NC; i++)
      if (tid + 256 * i < NJ * 64)
        lds_f[kWsConstsOff + tid + 256 * i] = cv[i];
#pragma unroll
    for (int i = 0; i < NX; i++)
      if (tid + 256 * i < a.n_xt * 256)
        lds_f[a.lds_xt_b / 4 + tid + 256 * i] = xv[i];
  }

  long long bar_cycles = 0, t_begin = 0; // PROF: cycles this wave spent inside barriers / total
  long long seg[5] = {0, 0, 0, 0, 0}; // PROF (compute waves): taps ready | conv done | x updated | published | job end
  long long t_seg = 0;
#define NAM_WS_STAMP(k, ...) \
  if constexpr (PROF) \
  { \
    asm volatile("" ::__VA_ARGS__); \
    const long long t_now = __builtin_readcyclecounter(); \
    seg[k] += t_now - t_seg; \
    t_seg = t_now; \
  }
  if constexpr (PROF)
    t_begin = __builtin_readcyclecounter();
  auto job_barrier = [&]() {
    if constexpr (PROF)
    {
      const long long t0 = __builtin_readcyclecounter();
      lds_barrier();
      bar_cycles += __builtin_readcyclecounter() - t0;
    }
    else
      lds_barrier();
  };
  if (w < 4)
  {
    // ------------------------------------------------ compute role ------------------------------------
    const int g = lane >> 4; // channel quad: this lane owns channels 4g..4g+3
    const int frame = 16 * w + (lane & 15);
    float* out = a.out ? a.out + (size_t)stream * a.io_stride : nullptr;
    const float head_scale = a.head_scale;
    const float act_p0 = a.act_p0;
    const unsigned v_g16 = (unsigned)g * 16u;
    const unsigned v_tap = (unsigned)(frame * SC) * 4u;
    const unsigned v_cond = lds_cond_b + (unsigned)frame * 4u;
    const unsigned v_lane16 = (unsigned)lane * 16u;
    const unsigned v_gh8 = (unsigned)(g & 1) * 16u + (unsigned)(g >> 1) * 8u; // half layout: the lane's channel pair
    // a job's operands: 4 tiles from tile buffer `tbuf`, its extra tile, 4 constant vectors
    auto load_ops = [&](Ops& o, const CDesc& J, int tbuf) {
      const unsigned a_t = v_lane16 + lds_tiles_b + (unsigned)tbuf * (kWsTileFloats * 4u);
#pragma unroll
      for (int q = 0; q < 4; q++)
        o.t[q] = lds_ld4(lds, a_t + 1024u * q);
      o.xt = lds_ld4(lds, v_lane16 + (unsigned)J.xt_b);
      const unsigned a_c = v_g16 + (unsigned)J.consts_b;
      o.bv = lds_ld4(lds, a_c);
      o.mv = lds_ld4(lds, a_c + 64u);
      o.b1v = lds_ld4(lds, a_c + 128u);
      o.ev = lds_ld4(lds, a_c + 192u);
    };
    f4 x = {0.f, 0.f, 0.f, 0.f}, head = {0.f, 0.f, 0.f, 0.f};
    int ji = 0, blk = 0;
    int nvalid = min(kBlock, a.n_frames);
    // descriptors: J (this job) <- Dn (next job: its operands are prefetched during this job) <- Dnn (the job after,
    // scalar-loaded in the shadow of this job's MFMAs so that no barrier's lgkmcnt(0) ever waits for it)
    CDesc Dn = P->cdesc[0];
    CDesc Dnn = P->cdesc[1];
    Ops ops[2];
    job_barrier(); // prologue barrier: consts, extra tiles, job 0's tiles / history / x0 are in LDS
    load_ops(ops[0], Dn, 0);

    for (int q0 = 0; q0 < total_pad; q0 += 2)
    {
#pragma unroll
      for (int u = 0; u < 2; u++)
      {
        const bool active = q0 + u < total;
        const CDesc J = Dn;
        Dn = Dnn;
        const int flags_rt = active ? J.flags : 0;
        const Ops& O = ops[u];
        job_barrier();
        if constexpr (PROF)
          t_seg = __builtin_readcyclecounter();
        const float cond = *reinterpret_cast<const float*>(lds + (v_cond + (unsigned)(blk & 1) * (kBlock * 4u)));
        // everything between the barrier and the publish, for NK k-steps per matrix (4: full layout, 2: half)
        // PLAIN: an ordinary layer (no array entry / exit work): the flag tests below fold away at compile time
        auto job_body = [&](auto nk_tag, auto plain_tag) {
          constexpr int NK = decltype(nk_tag)::value;
          constexpr bool PLAIN = decltype(plain_tag)::value;
          const int flags = PLAIN ? (int)CD_LAYER : flags_rt;
          // critical-path operand reads: the two shifted taps. Full layout: the lane's channel quad (16 B);
          // half layout: the two channels this lane feeds to the MFMAs (8 B).
          f4 bt0, bt1;
          if constexpr (NK == 4)
          {
            const unsigned a_tap = v_tap + min(v_g16, (unsigned)(J.gp & 0xff));
            bt0 = lds_ld4(lds, a_tap + (unsigned)J.tap0_b);
            bt1 = lds_ld4(lds, a_tap + (unsigned)J.tap1_b);
          }
          else
          {
            const f2 p0 = *reinterpret_cast<const f2*>(lds + (v_tap + v_gh8 + (unsigned)J.tap0_b));
            const f2 p1 = *reinterpret_cast<const f2*>(lds + (v_tap + v_gh8 + (unsigned)J.tap1_b));
            bt0 = f4{p0[0], p0[1], 0.f, 0.f};
            bt1 = f4{p1[0], p1[1], 0.f, 0.f};
          }
          NAM_WS_STAMP(0, "v"(bt0), "v"(bt1), "v"(cond))
          if (flags & CD_X0)
          {
            x = O.ev * cond; // ev = first array's rechannel column (in_size == 1)
            head = f4{0.f, 0.f, 0.f, 0.f};
          }
          else if (flags & CD_PRE_HEAD) // previous array's head rechannel + bias, in this array's layout
            head = ((flags & CD_PREV_HALF) ? mfma_n<2>(O.xt, head, f4{0.f, 0.f, 0.f, 0.f})
                                           : mfma_n<4>(O.xt, head, f4{0.f, 0.f, 0.f, 0.f}))
                   + O.ev;
          // dilated conv: 3 taps x NK k-steps. Tap 2 (the current frame) multiplies the lane's own x and needs
          // nothing from LDS, so its chain goes first and covers the latency of the two shifted-tap reads; the
          // conv bias and the input mixin ride in as initial accumulators. The next job's operands (its tiles
          // were dropped one job ago) are requested behind the taps, in the shadow of the MFMAs.
          load_ops(ops[u ^ 1], Dn, u ^ 1);
          {
            int jn = ji + 2;
            if (jn >= NJ)
              jn -= NJ;
            Dnn = P->cdesc[jn];
          }
          f4 acc0 = O.mv * cond, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = O.bv;
#pragma unroll
          for (int s = 0; s < NK; s++)
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(O.t[2][s], x[s], acc2, 0, 0, 0);
#pragma unroll
          for (int s = 0; s < NK; s++)
          {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(O.t[0][s], bt0[s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(O.t[1][s], bt1[s], acc1, 0, 0, 0);
          }
          const f4 pre = (acc0 + acc1) + acc2;
          NAM_WS_STAMP(1, "v"(pre))
          if (flags & CD_LAYER)
          {
            // half layout: only elements 0, 1 of a lane ever feed an MFMA (z into the 1x1, head into the head
            // rechannel), elements 2, 3 are the partner lane group's copies: two activations per lane, not four
            const f4 z = act4<ACT_T>(J.act, NK == 2 ? f4{pre[0], pre[1], pre[0], pre[1]} : pre, act_p0);
            head += z;
            // layer1x1 as two chains; the residual and the 1x1 bias are the initial accumulator
            f4 y0 = x + O.b1v, y1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NK; s += 2)
            {
              y0 = __builtin_amdgcn_mfma_f32_16x16x4f32(O.t[3][s], z[s], y0, 0, 0, 0);
              y1 = __builtin_amdgcn_mfma_f32_16x16x4f32(O.t[3][s + 1], z[s + 1], y1, 0, 0, 0);
            }
            x = y0 + y1;
            NAM_WS_STAMP(2, "v"(x))
            if (flags & CD_POST_OUT)
            {
              const f4 hout = mfma_n<NK>(O.xt, head, f4{0.f, 0.f, 0.f, 0.f}) + O.ev;
              if (out && g == 0 && frame < nvalid)
                out[(size_t)blk * kBlock + frame] = head_scale * hout[0];
            }
            else
            {
              if (flags & CD_POST_RECH)
                x = mfma_n<NK>(O.xt, x, f4{0.f, 0.f, 0.f, 0.f}); // next array's rechannel (no bias), its layout
              if (v_g16 <= (unsigned)(J.gp >> 8))
                lds_st4(lds, v_tap + v_g16 + (unsigned)J.pub_b, x);
            }
          }
          // What the job barrier waits for anyway, stated at the end of every variant: the four variants are
          // laid out one after the other behind flag tests, so without it the waitcnt pass carries one variant's
          // outstanding operand reads into the entry of the next and puts an lgkmcnt(0) in front of its first MFMA.
          // (the per-variant asm comment keeps the optimiser from sinking the four waits into one at the join.)
          __builtin_amdgcn_s_waitcnt(0xc07f);
          asm volatile("; end of job body nk=%0 plain=%1" ::"n"(NK), "n"((int)PLAIN));
        };
        const bool plain = (flags_rt & ~CD_HALF) == CD_LAYER;
        if (flags_rt & CD_HALF)
        {
          if (plain)
            job_body(std::integral_constant<int, 2>{}, std::true_type{});
          else
            job_body(std::integral_constant<int, 2>{}, std::false_type{});
        }
        else
        {
          if (plain)
            job_body(std::integral_constant<int, 4>{}, std::true_type{});
          else
            job_body(std::integral_constant<int, 4>{}, std::false_type{});
        }
        NAM_WS_STAMP(3, "v"(x))
        if (active && ++ji == NJ)
        {
          ji = 0;
          blk++;
          nvalid = min(kBlock, a.n_frames - blk * kBlock);
        }
        NAM_WS_STAMP(4, "s"(ji))
      }
    }
  }
  else
  {
    // ------------------------------------------------ mover role --------------------------------------
    const int mtid = tid - 256;
    const int hfr = 16 * (w - 4) + (lane >> 2); // frame inside a 64-frame set
    const unsigned v_hq16 = (unsigned)(lane & 3) * 16u; // channel quad
    const unsigned v_hist = (unsigned)(hfr * SC + 4 * (lane & 3)) * 4u;
    const unsigned v_mt16 = (unsigned)mtid * 16u;
    int* wpos_tbl = reinterpret_cast<int*>(st);
    const float* in = a.in ? a.in + (size_t)stream * a.io_stride : nullptr;
    const char* ibase = in ? reinterpret_cast<const char*>(in) : stb; // silence: any valid word, masked later
    const char* tiles0 = reinterpret_cast<const char*>(blob + a.tiles_off);
    int wposv = wpos_tbl[lane]; // lane r = write position of ring r
    const int ring_len_v = P->ring_len_by_id[lane];
    const f4 r1q = *reinterpret_cast<const f4*>(blob + a.r1_off + 4 * (lane & 3));

    // history of one job + the tiles of job `tjob`
    auto fetch = [&](HSlot& s, int f_rbase, int f_R, int f_LA, int f_LB, int f_ring_id, int f_q16max, bool next_block,
                     int jblk, int tjob) {
      int wp = __builtin_amdgcn_readlane(wposv, f_ring_id);
      if (next_block)
      {
        wp += kBlock;
        if (wp >= f_R)
          wp -= f_R;
      }
      const unsigned vq = min(v_hq16, (unsigned)f_q16max) + (unsigned)f_rbase;
      const unsigned cmul = (unsigned)f_q16max + 16u;
      const int Ls[2] = {f_LA, f_LB};
#pragma unroll
      for (int t = 0; t < 2; t++)
      {
        int sb = wp - Ls[t];
        if (sb < 0)
          sb += f_R;
        const unsigned v = (unsigned)(hfr + sb);
        const unsigned idx = min(v, v - (unsigned)f_R);
        // a job without a second set still issues the load (the number of loads in flight stays static),
        // but every lane reads the same cached 16 bytes
        const unsigned off = (t == 1 && f_LB == 0) ? 0u : __umul24(idx, cmul) + vq;
        s.h[t] = *reinterpret_cast<const f4*>(stb + off);
      }
      s.tile = *reinterpret_cast<const f4*>(tiles0 + ((unsigned)tjob * (kWsTileFloats * 4u) + v_mt16));
      int fi = jblk * kBlock + hfr;
      fi = min(fi, a.n_frames - 1);
      s.inp = *reinterpret_cast<const float*>(ibase + (in ? (unsigned)fi * 4u : 0u));
    };
    // drop a job's history (+ the following job's tiles) into LDS; for a block's first job also x0 and the inputs
    auto drop = [&](const HSlot& s, const VDesc& J, int succ_blk, int tbuf) {
      lds_st4(lds, v_hist + (unsigned)J.st_a_b, s.h[0]);
      if (J.flags & MV_SUCC_B)
        lds_st4(lds, v_hist + (unsigned)J.st_b_b, s.h[1]);
      lds_st4(lds, v_mt16 + lds_tiles_b + (unsigned)tbuf * (kWsTileFloats * 4u), s.tile);
      if (J.flags & MV_SUCC_FIRST)
      {
        const bool live = in && (succ_blk * kBlock + hfr < a.n_frames);
        const float iv = live ? s.inp : 0.0f;
        lds_st4(lds, v_hist + (unsigned)J.st_x0_b, r1q * iv);
        if ((lane & 3) == 0)
          *reinterpret_cast<float*>(lds + (lds_cond_b + (unsigned)((succ_blk & 1) * kBlock + hfr) * 4u)) = iv;
      }
    };

    HSlot slot[D];
    // job 0's tiles go straight to tile buffer 0; slot u = history of job u + tiles of job u + 1
    const f4 tile0 = *reinterpret_cast<const f4*>(tiles0 + v_mt16);
#pragma unroll
    for (int u = 0; u < D; u++)
    {
      const VDesc F = P->vdesc[u + NJ - 1 - D]; // the descriptor whose f_* fields describe job u
      fetch(slot[u], F.f_rbase, F.f_R, F.f_LA, F.f_LB, F.f_ring_id, F.f_q16max, false, 0, u + 1);
    }
    int ji = 0, blk = 0;
    int fj = D + 1, fblk = 0; // job / block whose history is fetched next
    int ftile = D + 2; // job whose tiles are fetched next (NJ >= D + 3)
    int nvalid = min(kBlock, a.n_frames);
    {
      // "job -1": job 0's history (and x0 / inputs of block 0) go to LDS, slot 0 is refilled with job D
      const VDesc J = P->vdesc[NJ - 1];
      lds_st4(lds, v_mt16 + lds_tiles_b, tile0);
      drop(slot[0], J, 0, 1);
      fetch(slot[0], J.f_rbase, J.f_R, J.f_LA, J.f_LB, J.f_ring_id, J.f_q16max, false, 0, D + 1);
    }
    VDesc Dn = P->vdesc[0];
    job_barrier(); // prologue barrier (matches the compute role)
    for (int q0 = 0; q0 < total_pad; q0 += D)
    {
#pragma unroll
      for (int u = 0; u < D; u++)
      {
        const bool active = q0 + u < total;
        const VDesc J = Dn;
        const int flags = active ? J.flags : 0;
        const int un = (u + 1) % D;
        job_barrier();
        Dn = P->vdesc[ji + 1 == NJ ? 0 : ji + 1]; // after the barrier: its lgkmcnt(0) must not wait for this load
        // this job's input rows (published by the previous job / dropped as x0) -> history ring
        if ((flags & MV_RING) && hfr < nvalid && v_hq16 <= (unsigned)J.q16max)
        {
          const f4 xin = lds_ld4(lds, v_hist + (unsigned)J.ap_src_b);
          const unsigned v = (unsigned)(__builtin_amdgcn_readlane(wposv, J.ring_id) + hfr);
          const unsigned widx = min(v, v - (unsigned)J.R);
          ring_store<WT>(stb, __umul24(widx, (unsigned)J.q16max + 16u) + v_hq16 + (unsigned)J.ring_b, xin);
        }
        // successor's history and the tiles of the job after it -> LDS (the halves of the double buffers
        // nobody reads during this job), then refill the slot (the other order — refill first — measured 6 % slower)
        drop(slot[un], J, blk + 1, (q0 + u) & 1);
        {
          const bool valid = fblk < n_blocks;
          fetch(slot[un], valid ? J.f_rbase : 0, valid ? J.f_R : 64, valid ? J.f_LA : 64, valid ? J.f_LB : 0,
                valid ? J.f_ring_id : 0, valid ? J.f_q16max : 0, valid && (fblk > blk), valid ? fblk : blk, ftile);
          if (++fj == NJ)
          {
            fj = 0;
            fblk++;
          }
          if (++ftile == NJ)
            ftile = 0;
        }
        if (active && ++ji == NJ)
        {
          ji = 0;
          wposv += nvalid;
          if (wposv >= ring_len_v)
            wposv -= ring_len_v;
          blk++;
          nvalid = min(kBlock, a.n_frames - blk * kBlock);
        }
      }
    }
    if (w == 4 && lane < a.n_rings)
      wpos_tbl[lane] = wposv;
  }
  if constexpr (PROF)
  {
    // rows 0..7 of the debug buffer: per wave of workgroup 0: {barrier cycles, total cycles}
    if (a.dbg && blockIdx.x == 0 && lane == 0)
    {
      a.dbg[w * 8 + 0] = bar_cycles;
      a.dbg[w * 8 + 1] = __builtin_readcyclecounter() - t_begin;
      for (int k = 0; k < 5; k++)
        a.dbg[w * 8 + 2 + k] = seg[k];
    }
  }
}

namespace
{
template <int ACT_T, bool WT, bool PROF, int D>
hipError_t launch_mfma_inst(const A1Args& a, int n_blocks, hipStream_t stream)
{
  static int lds_limit = 0; // per instantiation: dynamic LDS the runtime has been told about
  if (a.lds_bytes > lds_limit)
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&nam_a1_mfma_kernel<ACT_T, WT, PROF, D>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, a.lds_bytes);
    if (e != hipSuccess)
      return e;
    lds_limit = a.lds_bytes;
  }
  hipLaunchKernelGGL((nam_a1_mfma_kernel<ACT_T, WT, PROF, D>), dim3(n_blocks), dim3(512), a.lds_bytes, stream, a.plan,
                     a.blob, a);
  return hipGetLastError();
}
template <int ACT_T, bool WT, bool PROF>
hipError_t launch_mfma_depth(const A1Args& a, int n_blocks, hipStream_t stream)
{
  return a.prefetch == 5 ? launch_mfma_inst<ACT_T, WT, PROF, 5>(a, n_blocks, stream)
                         : launch_mfma_inst<ACT_T, WT, PROF, 6>(a, n_blocks, stream);
}
} // namespace

hipError_t launch_a1_mfma(const A1Args& a, int n_blocks, int act, hipStream_t stream)
{
  const bool wt = a.n_frames <= 2 * kBlock; // short launches write ring appends through (see ring_store)
  if (a.prefetch != 5 && a.prefetch != 6)
    return hipErrorInvalidValue;
  if (a.dbg) // developer tool: barrier-wait profile of workgroup 0
    return act == ACT_FASTTANH ? launch_mfma_depth<ACT_FASTTANH, false, true>(a, n_blocks, stream)
                               : launch_mfma_depth<-1, false, true>(a, n_blocks, stream);
  if (act == ACT_FASTTANH)
    return wt ? launch_mfma_depth<ACT_FASTTANH, true, false>(a, n_blocks, stream)
              : launch_mfma_depth<ACT_FASTTANH, false, false>(a, n_blocks, stream);
  if (act == ACT_TANH)
    return wt ? launch_mfma_depth<ACT_TANH, true, false>(a, n_blocks, stream)
              : launch_mfma_depth<ACT_TANH, false, false>(a, n_blocks, stream);
  return wt ? launch_mfma_depth<-1, true, false>(a, n_blocks, stream) : launch_mfma_depth<-1, false, false>(a, n_blocks, stream);
}

} // namespace namhip
