// kernel_a1_il.hip — nam_a1_il_kernel: the interleaved-frame fp32-MFMA kernel for the A1 family (kernel size 3).
#include "device_common.h"
#include "il_common.h"

namespace namhip
{

// ================================================================================================
// nam_a1_il_kernel — 4 compute waves + 1 loader wave per stream; compute wave w owns frames t = 4 j + w of the
// 64-frame block (j = lane & 15; lane group g = lane >> 4 owns a channel quad, exactly as in nam_a1_mfma_kernel:
// same A tiles, same constants, same "the lane's own values are its B operands" mapping — the MFMA does not care
// which frame a column stands for).
//
// Why interleave: a dilated tap of frame t is frame t - L. With 16 CONSECUTIVE frames per wave every layer needs
// frames of other waves (one LDS publish + workgroup barrier per layer: 20 per block for wavenet_a1_standard, each
// ~450 cycles of exposed latency on a lone wave per SIMD). With frames t = 4 j + w:
//   L multiple of 4 (d = 4, 8, 16, 32): t - L = 4 (j - L/4) + w — the SAME wave, L/4 lanes down the 16-lane row:
//                       one v_mov_dpp row_shr per value; lanes that fall off the row take the previous block's
//                       frame from the ring (requested long before)                                    [IL_DPP]
//   L >= 64          : the tap lies in an earlier block: the lane reads row (t - L) of the history ring in HBM
//                       itself, requested `D` jobs ahead                                               [IL_HIST]
//   otherwise (d = 1, 2, ...): other waves' frames: LDS window + one barrier                           [IL_EXCH]
// wavenet_a1_standard: 4 barriers per block instead of 20; 16 of 20 layers touch neither LDS (for activations) nor
// a barrier. No mover waves: each compute lane appends its own frame to the layer's ring (one 16-byte store per
// job) and issues its own two ring requests per job, D jobs ahead, into one of D fixed register slots consumed in
// request order (vmcnt retires in order). Lanes that need nothing request an out-of-range buffer offset (returns 0,
// no memory traffic), so the number of VMEM operations per job is static.
// The loader wave copies the constants, the extra tiles and every job's four weight tiles (4 KB) into LDS once per
// launch, job by job, and publishes its progress in an LDS word; compute waves read a job's operands from LDS into
// registers one job ahead (in the shadow of the MFMAs) and only look at the progress word they read alongside.
// State layout, rings and write positions are those of the other A1 kernels: interchangeable between launches.
// ================================================================================================


template <int ACT_T, bool WT, int D>
__global__ __launch_bounds__(320) void nam_a1_il_kernel(const A1Plan* __restrict__ P, const float* __restrict__ blob,
                                                        const A1Args a)
{
  using namespace mf;
  using il::kOob;
  using il::Ops;
  using il::Slot;
  extern __shared__ __attribute__((aligned(16))) float lds_il[];
  char* const lds = reinterpret_cast<char*>(lds_il);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = uni(tid >> 6);
  const int stream = a.stream_map ? a.stream_map[blockIdx.x] : (int)blockIdx.x;
  float* st = a.state + (size_t)stream * a.state_stride;
  const int NJ = a.il_jobs; // multiple of D
  const int n_blocks = (a.n_frames + kBlock - 1) / kBlock;
  const int total = n_blocks * NJ;
  // loader progress word: relaxed workgroup-scope atomics on an LDS-typed pointer = plain ds_read_b32 / ds_write_b32 that
  // the optimiser neither hoists out of the polling loop nor (as it does for a volatile access through the generic
  // `lds` pointer: a flat access) brackets with vmcnt(0) waits
  int* const progress = reinterpret_cast<int*>(lds_il) + a.il_flag_b / 4;
  auto progress_load = [&]() { return __hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };

  if (w == 4)
  {
    // ------------------------------------------------ loader role -------------------------------------
    if (lane == 0)
      __hip_atomic_store(progress, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    lds_barrier(); // nobody looks at the progress word before it is zeroed (LDS contents are undefined at launch)
    const f4* __restrict__ csrc = reinterpret_cast<const f4*>(blob + a.consts_off);
    const f4* __restrict__ xsrc = reinterpret_cast<const f4*>(blob + a.xt_off);
    const f4* __restrict__ tsrc = reinterpret_cast<const f4*>(blob + a.tiles_off);
    const int n_c4 = a.il_real_jobs * 16, n_x4 = a.n_xt * 64;
    for (int i = lane; i < n_c4; i += 64)
      lds_st4(lds, (unsigned)a.il_consts_b + (unsigned)i * 16u, csrc[i]);
    for (int i = lane; i < n_x4; i += 64)
      lds_st4(lds, (unsigned)a.il_xt_b + (unsigned)i * 16u, xsrc[i]);
    // tiles: 4 jobs (16 KB) requested at a time; a job is published as soon as its 4 KB are in LDS
    constexpr int kB = 4;
    for (int j0 = 0; j0 < a.il_real_jobs; j0 += kB)
    {
      f4 v[kB][4];
#pragma unroll
      for (int jj = 0; jj < kB; jj++)
#pragma unroll
        for (int q = 0; q < 4; q++)
          v[jj][q] = tsrc[(size_t)min(j0 + jj, a.il_real_jobs - 1) * 256 + q * 64 + lane];
#pragma unroll
      for (int jj = 0; jj < kB; jj++)
        if (j0 + jj < a.il_real_jobs)
        {
#pragma unroll
          for (int q = 0; q < 4; q++)
            lds_st4(lds, (unsigned)a.il_tiles_b + (unsigned)(j0 + jj) * 4096u + (unsigned)q * 1024u + (unsigned)lane * 16u,
                    v[jj][q]);
          __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): the tiles (and, the first time, constants) are in LDS
          if (lane == 0)
            __hip_atomic_store(progress, j0 + jj + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    // the compute waves' workgroup barriers count every wave of the workgroup
    const int n_bar = n_blocks * a.il_exch; // (+ the one above)
    for (int i = 0; i < n_bar; i++)
      lds_barrier();
    return;
  }

  // -------------------------------------------------- compute role ------------------------------------
  const int g = lane >> 4; // channel quad
  const int j = lane & 15;
  const int t = 4 * j + w; // this lane's frame inside the block
  const float* in = a.in ? a.in + (size_t)stream * a.io_stride : nullptr;
  float* out = a.out ? a.out + (size_t)stream * a.io_stride : nullptr;
  const float head_scale = a.head_scale;
  const float act_p0 = a.act_p0;
  const unsigned v_g16 = (unsigned)g * 16u;
  const unsigned v_gh8 = (unsigned)(g & 1) * 16u + (unsigned)(g >> 1) * 8u; // half layout: the lane's channel pair
  const unsigned v_lane16 = (unsigned)lane * 16u;
  const bool hi_pair = (g >> 1) != 0; // half layout: this lane's pair is elements 2, 3 of its 16-byte quad
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)st, 0, (int)(a.state_stride * 4), 0x00020000);
  const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc((void*)(in ? in : st), 0, in ? a.n_frames * 4 : 0, 0x00020000);
  const auto rsrc_out = __builtin_amdgcn_make_buffer_rsrc((void*)(out ? out : st), 0, out ? a.n_frames * 4 : 0, 0x00020000);
  int* wpos_tbl = reinterpret_cast<int*>(st);
  int wposv = lane < a.n_rings ? wpos_tbl[lane] : 0; // lane r = write position of ring r
  const int ring_len_v = P->ring_len_by_id[lane];
  int ji = 0, blk = 0; // job inside the block / block inside the launch

  // the two ring requests of the job described by F (`ahead` = 1: the job belongs to the next block)
  auto fetch = [&](Slot& s, const IlFetch& F, int ahead, bool valid) {
    int wp = __builtin_amdgcn_readlane(wposv, F.ring_id) + (ahead ? kBlock : 0);
    if (wp >= F.R)
      wp -= F.R;
    const bool half = F.row_b == 32;
    const unsigned chan = min(half ? (unsigned)(g & 1) * 16u : v_g16, (unsigned)F.row_b - 16u);
    const unsigned base = (unsigned)F.ring_b + chan;
#pragma unroll
    for (int q = 0; q < 2; q++)
    {
      const int L = q == 0 ? F.LA : F.LB;
      const int n = q == 0 ? F.nA : F.nB;
      // row (t - L) lies before the block for every requesting lane: index (wp + t - L) mod R, computed without a sign
      const unsigned v = (unsigned)(wp + t - L + F.R);
      const unsigned idx = min(v, v - (unsigned)F.R);
      const bool want = valid && L > 0 && j < n;
      const unsigned off = want ? __umul24(idx, (unsigned)F.row_b) + base : kOob;
      const f4 r = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0));
      if (q == 0)
        s.a = r;
      else
        s.b = r;
    }
    s.inp = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_in, t * 4, uni((blk + ahead) * (kBlock * 4)), 0));
  };
  // a job's operands from LDS (offsets given by the PREVIOUS job's descriptor), progress word first
  auto load_ops = [&](Ops& o, int consts_b, int xt_b, int tiles_b) {
    o.ready = progress_load();
    const unsigned a_t = v_lane16 + (unsigned)tiles_b;
#pragma unroll
    for (int q = 0; q < 4; q++)
      o.t[q] = lds_ld4(lds, a_t + 1024u * q);
    o.xt = lds_ld4(lds, v_lane16 + (unsigned)xt_b);
    const unsigned a_c = v_g16 + (unsigned)consts_b;
    o.bv = lds_ld4(lds, a_c);
    o.mv = lds_ld4(lds, a_c + 64u);
    o.b1v = lds_ld4(lds, a_c + 128u);
    o.ev = lds_ld4(lds, a_c + 192u);
  };
  // wait (bounded) until the loader has published `need` jobs
  auto wait_loader = [&](int need) {
    int spins = 0;
    while (uni(progress_load()) < need)
    {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1 << 22))
        __builtin_trap(); // never a silent hang: the loader wave cannot be blocked by anything
    }
  };

  Slot slot[D];
#pragma unroll
  for (int u = 0; u < D; u++)
  {
    // (the same five-operation pattern as a job of the main loop, so that the waitcnt pass sees one request order on
    // the loop's entry and back edges)
    __builtin_amdgcn_raw_buffer_store_b128(il::u4{0u, 0u, 0u, 0u}, rsrc, (int)kOob, 0, WT ? 17 : 0);
    fetch(slot[u], P->il_fetch[NJ - D + u], 0, true); // position NJ - D + u describes job u
    __builtin_amdgcn_raw_buffer_store_b32(0u, rsrc_out, (int)kOob, 0, 0);
  }
  lds_barrier(); // matches the loader's: the progress word is zeroed (the requests above are already in flight)

  f4 x = {0.f, 0.f, 0.f, 0.f}, head = {0.f, 0.f, 0.f, 0.f};
  int nvalid = min(kBlock, a.n_frames);
  unsigned win_par = 0; // which LDS window the next IL_EXCH job uses
  IlDesc Dn = P->il_desc[0];
  Ops ops[2];
  {
    const IlDesc last = P->il_desc[NJ - 1]; // its "next job" fields describe job 0
    wait_loader(last.n_ready);
    load_ops(ops[0], last.n_consts_b, last.n_xt_b, last.n_tiles_b);
  }
  // the loop is unrolled kU jobs deep (the jobs per block are a multiple of it): request slot u % D and operand
  // buffer u & 1 are compile-time register sets
  constexpr int kU = 10;
  static_assert(kU % D == 0 && kU % 2 == 0, "slot / operand indices must come back to 0 at the top of the unrolled body");

  for (int q0 = 0; q0 < total; q0 += kU)
  {
#pragma unroll
    for (int u = 0; u < kU; u++)
    {
      const IlDesc J = Dn;
      Dn = P->il_desc[ji + 1 == NJ ? 0 : ji + 1];
      const IlFetch F = P->il_fetch[ji];
      const int flags = J.flags;
      const bool real = J.kind != IL_IDLE;
      // nothing moves across a job boundary (left alone, the scheduler hoists the next jobs' LDS / ring requests far up
      // and the live ranges no longer fit the register file)
      __builtin_amdgcn_sched_barrier(0);
      // Every job issues the same five vector-memory operations in straight-line code — append, two ring requests,
      // input sample, output sample — with out-of-range offsets where there is nothing to do. A VMEM operation inside
      // a branch (or a different count on two paths) makes hipcc's waitcnt pass forget the request order, and every
      // wait behind it becomes vmcnt(0).
      const Slot S = slot[u % D];
      asm volatile("" ::"v"(S.a), "v"(S.b), "v"(S.inp)); // one wait for the whole slot (the oldest requests in flight)
      const float cond = S.inp;
      Ops& O = ops[u & 1];
      if (real && uni(O.ready) < ji + 1) // the loader had not published this job when its operands were read
      {
        wait_loader(ji + 1);
        const IlDesc Pv = P->il_desc[ji == 0 ? NJ - 1 : ji - 1]; // (padding jobs pass job 0's operands along)
        load_ops(O, Pv.n_consts_b, Pv.n_xt_b, Pv.n_tiles_b);
      }
      const unsigned g16max = (unsigned)J.gp;
      if (flags & CD_X0)
      {
        x = O.ev * cond; // ev = first array's rechannel column (in_size == 1)
        head = f4{0.f, 0.f, 0.f, 0.f};
      }
      // this job's input -> its history ring (row of frame t), 16 bytes per lane that owns a channel quad
      {
        const unsigned v = (unsigned)(__builtin_amdgcn_readlane(wposv, J.ring_id) + t);
        const unsigned widx = min(v, v - (unsigned)J.R);
        const bool ok = real && t < nvalid && v_g16 <= g16max;
        const unsigned off = ok ? __umul24(widx, (unsigned)J.row_b) + v_g16 + (unsigned)J.ring_b : kOob;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(il::u4, x), rsrc, (int)off, 0, WT ? 17 : 0);
      }
      // the slot is consumed (copied) above: request the job D ahead into it, then the next job's operands
      {
        const int ahead = ji + D >= NJ ? 1 : 0;
        fetch(slot[u % D], F, ahead, !ahead || blk + 1 < n_blocks);
      }
      load_ops(ops[(u + 1) & 1], J.n_consts_b, J.n_xt_b, J.n_tiles_b);
      float yout = 0.0f;
      if (real)
      {
        auto job_body = [&](auto nk_tag) {
          constexpr int NK = decltype(nk_tag)::value;
          // the lane's slice of a raw ring row: its quad (full layout) or its pair inside the quad (half layout)
          auto slice = [&](const f4& r) { return NK == 4 ? r : (hi_pair ? f4{r[2], r[3], 0.f, 0.f} : f4{r[0], r[1], 0.f, 0.f}); };
          f4 bt0, bt1;
          if (J.kind == IL_HIST)
          {
            bt0 = slice(S.a);
            bt1 = slice(S.b);
          }
          else if (J.kind == IL_DPP)
          {
            const f4 pa = slice(S.a), pb = slice(S.b);
            bt0 = bt1 = f4{0.f, 0.f, 0.f, 0.f};
            switch (J.dil)
            {
              case 4: il::dpp_taps<NK, 1>(x, pa, pb, bt0, bt1); break;
              case 8: il::dpp_taps<NK, 2>(x, pa, pb, bt0, bt1); break;
              case 16: il::dpp_taps<NK, 4>(x, pa, pb, bt0, bt1); break;
              default: il::dpp_taps<NK, 8>(x, pa, pb, bt0, bt1); break;
            }
          }
          else
          {
            // IL_EXCH: every wave publishes its frames (and the same frames of the previous block) to the window,
            // one workgroup barrier, then each lane reads the two shifted rows
            // (rows in (frame & 3, frame >> 2) order, 80 bytes each: the wavefront's lanes hold frames 4 apart, and in
            // frame order the sixteen lanes of a b128 phase shared two bank groups — see kernel_a1_p2.hip)
            constexpr unsigned kRowB = 80u;
            static_assert(2 * kBlock * kRowB <= (unsigned)kIlWinB, "exchange window");
            auto win_off = [](unsigned F) { return ((F & 64u) + ((F & 3u) << 4) + ((F & 63u) >> 2)) * kRowB; };
            const unsigned wb = win_par * (unsigned)kIlWinB;
            win_par ^= 1u;
            if (v_g16 <= g16max)
            {
              lds_st4(lds, wb + win_off((unsigned)(kBlock + t)) + v_g16, x);
              lds_st4(lds, wb + win_off((unsigned)t) + v_g16, S.a);
            }
            lds_barrier();
            const unsigned chan = NK == 4 ? min(v_g16, g16max) : v_gh8;
            const unsigned r1 = wb + win_off((unsigned)(kBlock + t - J.dil)) + chan;
            const unsigned r0 = wb + win_off((unsigned)(kBlock + t - (J.tap0_lds ? 2 * J.dil : 0))) + chan;
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
            if (!J.tap0_lds)
              bt0 = slice(S.b);
          }
          if (flags & CD_PRE_HEAD) // previous array's head rechannel + bias, in this array's layout
            head = ((flags & CD_PREV_HALF) ? mfma_n<2>(O.xt, head, f4{0.f, 0.f, 0.f, 0.f})
                                           : mfma_n<4>(O.xt, head, f4{0.f, 0.f, 0.f, 0.f}))
                   + O.ev;
          // dilated conv: 3 taps x NK k-steps, three independent chains; bias and input mixin ride in as accumulators
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
          const f4 z = act4<ACT_T>(J.act, NK == 2 ? f4{pre[0], pre[1], pre[0], pre[1]} : pre, act_p0);
          head += z;
          // pin the accumulator: otherwise the optimiser sinks these adds to the next USE of `head` (the array's last
          // job) and keeps every job's z alive until then
          asm volatile("" : "+v"(head));
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
          if (flags & CD_POST_OUT)
            yout = head_scale * (mfma_n<NK>(O.xt, head, f4{0.f, 0.f, 0.f, 0.f}) + O.ev)[0];
          else if (flags & CD_POST_RECH)
            x = mfma_n<NK>(O.xt, x, f4{0.f, 0.f, 0.f, 0.f}); // next array's rechannel (no bias), its layout
        };
        if (flags & CD_HALF)
          job_body(std::integral_constant<int, 2>{});
        else
          job_body(std::integral_constant<int, 4>{});
      }
      // the block's output sample (offset out of range on every job but the block's last)
      {
        const bool ok = (flags & CD_POST_OUT) && g == 0 && t < nvalid;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yout), rsrc_out, ok ? t * 4 : (int)kOob,
                                              uni(blk * (kBlock * 4)), 0);
      }
      if (++ji == NJ)
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
  if (w == 0 && lane < a.n_rings)
    wpos_tbl[lane] = wposv;
}

namespace
{
template <int ACT_T, bool WT, int D>
hipError_t launch_il_inst(const A1Args& a, int n_blocks, hipStream_t stream)
{
  static int lds_limit = 0;
  if (a.il_lds_bytes > lds_limit)
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&nam_a1_il_kernel<ACT_T, WT, D>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, a.il_lds_bytes);
    if (e != hipSuccess)
      return e;
    lds_limit = a.il_lds_bytes;
  }
  hipLaunchKernelGGL((nam_a1_il_kernel<ACT_T, WT, D>), dim3(n_blocks), dim3(320), a.il_lds_bytes, stream, a.plan, a.blob, a);
  return hipGetLastError();
}
template <int ACT_T, bool WT>
hipError_t launch_il_depth(const A1Args& a, int n_blocks, hipStream_t stream)
{
  return a.il_depth == 10 ? launch_il_inst<ACT_T, WT, 10>(a, n_blocks, stream) : launch_il_inst<ACT_T, WT, 5>(a, n_blocks, stream);
}
} // namespace

hipError_t launch_a1_il(const A1Args& a, int n_blocks, int act, hipStream_t stream)
{
  const bool wt = a.n_frames <= 2 * kBlock; // short launches write ring appends through (see ring_store)
  if (a.il_depth != 5 && a.il_depth != 10)
    return hipErrorInvalidValue;
  if (act == ACT_FASTTANH)
    return wt ? launch_il_depth<ACT_FASTTANH, true>(a, n_blocks, stream) : launch_il_depth<ACT_FASTTANH, false>(a, n_blocks, stream);
  if (act == ACT_TANH)
    return wt ? launch_il_depth<ACT_TANH, true>(a, n_blocks, stream) : launch_il_depth<ACT_TANH, false>(a, n_blocks, stream);
  return wt ? launch_il_depth<-1, true>(a, n_blocks, stream) : launch_il_depth<-1, false>(a, n_blocks, stream);
}

} // namespace namhip
