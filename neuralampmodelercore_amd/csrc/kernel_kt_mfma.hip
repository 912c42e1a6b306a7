// kernel_kt_mfma.hip — nam_kt_mfma_kernel: fp32 MFMA for single-array models with any per-layer kernel size (A2).
#include "device_common.h"

namespace namhip
{

// ------------------------------------------------------------------------------------------------
// K-tap MFMA kernel (A2 family: single layer array, per-layer kernel sizes up to 16, head rechannel with taps)
// ------------------------------------------------------------------------------------------------
// One workgroup of four wavefronts per stream, wave w owns frames [16w, 16w + 16) of the 64-frame block, lane
// layout and MFMA operand mapping exactly as in nam_a1_mfma_kernel (full layout, or half layout for C = 8).
// A layer is a run of CHUNKS of up to kKtTaps taps (plan.h: KtDesc). Every tap's B operand is the lane's slice of
// frame (t - L): from the LDS copy of the layer input when that frame is inside the block, else from the layer's
// history ring in HBM — requested D CHUNKS AHEAD with buffer loads whose offset is out of range for lanes that do
// not need history (no memory traffic), together with that chunk's tap tiles. Only a layer's
// last chunk activates, runs the 1x1, publishes and meets the barrier: one barrier per layer. The head rechannel
// (A2: 16 taps over the head accumulator) is one more layer whose published input is the head accumulator.
// State layout, ring geometry and write positions are nam_a1_kernel's (the two are interchangeable mid-stream).
template <int NK, bool WT, int ACT_T>
__global__ __launch_bounds__(256) void nam_kt_mfma_kernel(const A1Plan* __restrict__ P, const float* __restrict__ blob,
                                                          const A1Args a)
{
  using mf::f2;
  using mf::f4;
  using fN = std::conditional_t<NK == 2, f2, f4>;
  using uN = std::conditional_t<NK == 2, __attribute__((ext_vector_type(2))) unsigned,
                                __attribute__((ext_vector_type(4))) unsigned>;
  // Everything a chunk needs from memory — history operands (a ring row written by an earlier launch comes from HBM:
  // 2-3 us), its tap tiles, the next block's input sample — is requested D chunks ahead, into one of D fixed
  // register sets, and consumed in request order: vmcnt retires in order, so a wait for anything younger would also
  // wait for every older request. The chunk loop is unrolled D times (one body per set); per chunk 1 store +
  // kOps loads, D * (kOps + 1) <= 63 outstanding.
  constexpr int D = NK == 2 ? 5 : 3;
  constexpr int NT = NK == 2 ? kKtTaps / 2 : kKtTaps; // 16-byte tile loads per chunk (half layout: two taps each)
  extern __shared__ __attribute__((aligned(16))) float lds_kt[];
  char* const lds = reinterpret_cast<char*>(lds_kt);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = uni(tid >> 6);
  const int g = lane >> 4;
  const int frame = 16 * w + (lane & 15);
  const int stream = a.stream_map ? a.stream_map[blockIdx.x] : (int)blockIdx.x;
  float* st = a.state + (size_t)stream * a.state_stride;
  const float* in = a.in ? a.in + (size_t)stream * a.io_stride : nullptr;
  float* out = a.out ? a.out + (size_t)stream * a.io_stride : nullptr;
  const int C = P->arr[0].channels;
  const int act = P->arr[0].act;
  const float act_p0 = a.act_p0, head_scale = a.head_scale;
  const int NCH = P->kt_chunks;
  const int n_blocks = (a.n_frames + kBlock - 1) / kBlock;
  const int total = n_blocks * NCH;
  const int total_pad = (total + D - 1) / D * D;
  const unsigned row_b = (unsigned)(C + 4) * 4u; // LDS row pitch of a published frame
  const unsigned buf_b = (kBlock + 1) * row_b; // two buffers, alternating per layer; row 0 of each is zero: a tap whose
                                               // frame lies before the block reads it and takes its operand from history
  const unsigned aux_b = 2u * buf_b; // LDS copy of the 1x1 tiles and the constants
  const unsigned ring_row_b = (unsigned)C * 4u;
  // the lane's B-operand slice of a frame row (its own channels, see nam_a1_mfma_kernel) and the quad it publishes
  const unsigned opnd_b = NK == 2 ? (unsigned)((g & 1) * 16 + (g >> 1) * 8) : min((unsigned)g * 16u, ring_row_b - 16u);
  const bool pub_lane = NK == 2 ? g < 2 : 4 * g < C;
  const unsigned quad_b = (unsigned)g * 16u;
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)st, 0, (int)(a.state_stride * 4), 0x00020000);
  // tiles: scalar base + scalar offset per load, the lane only contributes lane * 16; input samples: reads beyond the
  // launch's frames (or without an input) return 0
  const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)blob, 0, 0x7fffffff, 0x00020000);
  const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc((void*)(in ? in : st), 0, in ? a.n_frames * 4 : 0, 0x00020000);
  const int lane16 = lane * 16, frame4 = frame * 4;
  constexpr unsigned kOob = 0x7ffffff0u; // beyond num_records: the load returns 0 without touching memory
  int* wpos_tbl = reinterpret_cast<int*>(st);
  int wposv = lane < a.n_rings ? wpos_tbl[lane] : 0; // lane r = write position of ring r
  const int ring_len_v = P->ring_len_by_id[lane];
  const f4 rech = *reinterpret_cast<const f4*>(blob + P->kt_rech_off + g * 4);
  {
    const f4* __restrict__ src = reinterpret_cast<const f4*>(blob + P->kt_lds_src_off);
    const int n4 = P->kt_lds_floats / 4;
    for (int i = tid; i < n4; i += 256)
      *reinterpret_cast<f4*>(lds + aux_b + (unsigned)i * 16u) = src[i];
  }

  int blk = 0, ci = 0;
  struct Set
  {
    fN th[kKtTaps]; // history operands (lanes whose frame - L lies before the block)
    f4 tt[NT]; // tap tiles
    float cn; // input sample of the block after the chunk's
  };
  Set S[D];
  // operands of the chunk described by Dq, which lies `ahead` (0 / 1) blocks after the current one
  auto fetch = [&](Set& s, const KtDesc& Dq, int ahead) {
    int wp = __builtin_amdgcn_readlane(wposv, Dq.ring_id) + (ahead ? kBlock : 0);
    if (wp >= Dq.R)
      wp -= Dq.R;
#pragma unroll
    for (int i = 0; i < kKtTaps; i++)
    {
      // branch-free: lanes whose frame is inside the block (and taps the chunk does not have) get an offset beyond
      // the buffer; the rest row (wp + frame - L) mod R of the ring
      const int tl = frame - (i < Dq.ntaps ? Dq.L[i] : 0);
      int idx = wp + tl;
      idx += (idx >> 31) & Dq.R;
      const unsigned real = __umul24((unsigned)idx, ring_row_b) + ((unsigned)Dq.ring_b + opnd_b);
      const unsigned hmask = (unsigned)(tl >> 31); // all ones: history
      const unsigned off = (real & hmask) | (kOob & ~hmask);
      uN raw;
      if constexpr (NK == 2)
        raw = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)off, 0, 0);
      else
        raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0);
      s.th[i] = __builtin_bit_cast(fN, raw);
    }
#pragma unroll
    for (int i = 0; i < NT; i++)
      s.tt[i] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, lane16, Dq.tile_off * 4 + i * 1024, 0));
    s.cn = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_in, frame4, uni((blk + ahead + 1) * (kBlock * 4)), 0));
  };
  auto wrap = [&](int c) { return c >= NCH ? c - NCH : c; }; // NCH > D (plan_a1.cpp)

  int nvalid = min(kBlock, a.n_frames);
  unsigned par = 0;
  float cond = (in && frame < nvalid) ? in[frame] : 0.0f;
  f4 x = rech * cond, head = {0.f, 0.f, 0.f, 0.f};
  f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  f4 pend = x; // rows published last (in LDS buffer `par`): appended to the consuming layer's ring by its first chunk
  if (tid < 64 && (unsigned)(tid & 31) * 4u < row_b)
    *reinterpret_cast<float*>(lds + ((unsigned)(tid >> 5) * buf_b + (unsigned)(tid & 31) * 4u)) = 0.0f;
  if (pub_lane)
    mf::lds_st4(lds, (unsigned)(frame + 1) * row_b + quad_b, x);
  // prologue: the operands of chunks 0 .. D-1
#pragma unroll
  for (int c = 0; c < D; c++)
  {
    const KtDesc Dq = P->kt_desc[c];
    fetch(S[c], Dq, 0);
  }
  KtDesc Dn = P->kt_desc[0];
  mf::lds_barrier();

  for (int q0 = 0; q0 < total_pad; q0 += D)
  {
#pragma unroll
    for (int u = 0; u < D; u++)
    {
      const bool active = q0 + u < total;
      const KtDesc J = Dn;
      const int flags = active ? J.flags : 0;
      const int ntaps = active ? J.ntaps : 0;
      Set& s = S[u];
      Dn = P->kt_desc[wrap(ci + 1)];
      const KtDesc Dq = P->kt_desc[wrap(ci + D)]; // the chunk this set is refilled for
      // in-block operands of this chunk's taps (rows of the layer input published before the last barrier), the
      // layer's constants and 1x1 tile
      fN lv[kKtTaps];
#pragma unroll
      for (int i = 0; i < kKtTaps; i++)
      {
        const int row = max(frame + 1 - J.L[i], 0);
        lv[i] = *reinterpret_cast<const fN*>(lds + (__umul24((unsigned)row, row_b) + (par * buf_b + opnd_b)));
      }
      const unsigned a_c = aux_b + (unsigned)J.consts_off + quad_b;
      const f4 bv = mf::lds_ld4(lds, a_c), mv = mf::lds_ld4(lds, a_c + 64u), b1v = mf::lds_ld4(lds, a_c + 128u);
      const fN w1 = *reinterpret_cast<const fN*>(lds + (aux_b + (unsigned)J.w1_off + (unsigned)lane * (NK * 4u)));
      // a layer's first chunk appends the layer input (still in `pend`) to the layer's ring
      {
        int v = __builtin_amdgcn_readlane(wposv, J.ring_id) + frame;
        if (v >= J.R)
          v -= J.R;
        const bool app = (flags & KT_FIRST) && (flags & KT_RING) && pub_lane && frame < nvalid;
        const unsigned off = app ? (unsigned)J.ring_b + (unsigned)v * ring_row_b + quad_b : kOob;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, pend),
                                               rsrc, (int)off, 0, WT ? /*sc0 sc1*/ 17 : 0);
      }
      if (flags & KT_FIRST)
      {
        acc0 = bv + mv * cond;
        acc1 = f4{0.f, 0.f, 0.f, 0.f};
      }
      // one wait for the whole set (the oldest requests in flight), not one per tap
#pragma unroll
      for (int i = 0; i < kKtTaps; i++)
        asm volatile("" ::"v"(s.th[i]));
#pragma unroll
      for (int i = 0; i < NT; i++)
        asm volatile("" ::"v"(s.tt[i]));
      auto tap = [&](int i) {
        const fN bsum = lv[i] + s.th[i]; // exactly one of the two is the operand, the other is 0
#pragma unroll
        for (int m = 0; m < NK; m++)
        {
          const float bm = bsum[m];
          const float am = NK == 2 ? s.tt[i / 2][(i % 2) * 2 + m] : s.tt[i][m];
          if (i & 1)
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(am, bm, acc1, 0, 0, 0);
          else
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(am, bm, acc0, 0, 0, 0);
        }
      };
      if (ntaps == kKtTaps) // the common case (A2: 26 of 30 chunks) without a test per tap
      {
#pragma unroll
        for (int i = 0; i < kKtTaps; i++)
          tap(i);
      }
      else
      {
#pragma unroll
        for (int i = 0; i < kKtTaps; i++)
          if (i < ntaps)
            tap(i);
      }
      const float cn = s.cn;
      // refill this set for chunk ci + D (beyond the end of this block it belongs to the next one, whose rings have
      // moved on by one block)
      fetch(s, Dq, ci + D >= NCH ? 1 : 0);
      if (flags & KT_LAST)
      {
        const f4 pre = acc0 + acc1;
        f4 pub;
        if (flags & KT_HEAD)
        {
          if (out && g == 0 && frame < nvalid)
            out[(size_t)blk * kBlock + frame] = head_scale * pre[0];
          // block boundary: ring write positions move on, the next block's layer 0 input is rechannel * sample
          wposv += nvalid;
          if (wposv >= ring_len_v)
            wposv -= ring_len_v;
          blk++;
          nvalid = min(kBlock, a.n_frames - blk * kBlock);
          cond = cn;
          x = rech * cond;
          head = f4{0.f, 0.f, 0.f, 0.f};
          pub = x;
        }
        else
        {
          const f4 z = mf::act4<ACT_T>(act, pre, act_p0);
          head += z;
          f4 y0 = x + b1v, y1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int m = 0; m < NK; m += 2)
          {
            y0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[m], z[m], y0, 0, 0, 0);
            y1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[m + 1], z[m + 1], y1, 0, 0, 0);
          }
          x = y0 + y1;
          pub = (flags & KT_NEXT_HEAD) ? head : x;
        }
        par ^= 1u;
        if (pub_lane)
          mf::lds_st4(lds, par * buf_b + (unsigned)(frame + 1) * row_b + quad_b, pub);
        pend = pub;
        mf::lds_barrier();
      }
      if (active && ++ci == NCH)
        ci = 0;
    }
  }
  if (w == 0 && lane < a.n_rings)
    wpos_tbl[lane] = wposv;
}

hipError_t launch_kt_mfma(const A1Args& a, int n_blocks, int nk, int channels, int lds_aux_floats, int act,
                          hipStream_t stream)
{
  const bool wt = a.n_frames <= 2 * kBlock; // short launches write ring appends through (see ring_store)
  const int lds_bytes = (2 * (kBlock + 1) * (channels + 4) + lds_aux_floats) * (int)sizeof(float);
  if (lds_bytes > 64 * 1024)
    return hipErrorInvalidValue; // (plan_a1.cpp keeps the LDS copy small; 16-channel models with 32 layers stay below)
#define NAM_KT(NK, WT, ACT) \
  hipLaunchKernelGGL((nam_kt_mfma_kernel<NK, WT, ACT>), dim3(n_blocks), dim3(256), lds_bytes, stream, a.plan, a.blob, a)
#define NAM_KT_ACT(NK, WT) \
  switch (act) \
  { \
    case ACT_LEAKYRELU: NAM_KT(NK, WT, ACT_LEAKYRELU); break; \
    case ACT_FASTTANH: NAM_KT(NK, WT, ACT_FASTTANH); break; \
    case ACT_TANH: NAM_KT(NK, WT, ACT_TANH); break; \
    default: NAM_KT(NK, WT, -1); break; \
  }
  if (nk == 2)
  {
    if (wt)
      NAM_KT_ACT(2, true)
    else
      NAM_KT_ACT(2, false)
  }
  else
  {
    if (wt)
      NAM_KT_ACT(4, true)
    else
      NAM_KT_ACT(4, false)
  }
#undef NAM_KT_ACT
#undef NAM_KT
  return hipGetLastError();
}

} // namespace namhip
