// plan_a1.cpp — the A1-family kernels' plans (nam_a1_kernel, nam_a1_mfma_kernel, nam_a1_p2 / p4 / q, nam_kt_mfma_kernel,
// nam_kq_kernel): eligibility, job descriptors, lane-record tiles, LDS layouts, the plan-time channel padding. See plan_internal.h.
#include "plan_internal.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <sstream>

namespace namhip
{
// --------------------------------------------------------------------------------------------
// A1-family fast path eligibility + packing
// --------------------------------------------------------------------------------------------
bool a1_channel_supported(int c)
{
  return c == 1 || c == 2 || c == 3 || c == 4 || c == 6 || c == 8 || c == 12 || c == 16;
}

// Job table of nam_a1_mfma_kernel (plan.h: CDesc / VDesc). Requires channels % 4 == 0 (<= 16), kernel size 3,
// a mono input, and at least two layers per array (the extra tile of a job serves either its array's entry
// or its exit).
void build_a1_ws(const WaveNetSpec& wn, Plan& plan)
{
  A1Plan& a1 = plan.a1;
  const int n_arrays = (int)wn.arrays.size();
  int n_layers = 0;
  for (const auto& A : wn.arrays)
  {
    if (A.num_layers() < 2)
      return;
    n_layers += A.num_layers();
  }
  if (wn.arrays[0].input_size != 1)
    return;
  for (size_t ai = 0; ai < wn.arrays.size(); ai++)
  {
    const LayerArraySpec& A = wn.arrays[ai];
    if (A.channels % 4 != 0 || A.channels > 16 || A.head_size > 16 || A.head_kernel_size != 1)
      return;
    for (int k : A.kernel_sizes)
      if (k != 3)
        return;
    if (ai > 0 && (A.input_size % 4 != 0 || A.input_size > 16))
      return;
  }
  const int NJ = (n_layers + 1) / 2 * 2;
  const int n_xt = 2 * n_arrays - 1;
  const int PF = (NJ % 5 == 0) ? 5 : 6; // mover prefetch depth (plan.h)
  if (NJ > kWsJobMax || NJ < PF + 3 || n_xt > kWsXtMax)
    return;
  a1.ws_prefetch = PF;
  while (plan.blob.size() % 64)
    plan.blob.push_back(0.0f);
  a1.ws_tiles_off = (int)plan.blob.size();
  plan.blob.resize(plan.blob.size() + (size_t)NJ * kWsTileFloats, 0.0f);
  a1.ws_xt_off = (int)plan.blob.size();
  plan.blob.resize(plan.blob.size() + (size_t)kWsXtMax * 256, 0.0f);
  a1.ws_consts_off = (int)plan.blob.size();
  plan.blob.resize(plan.blob.size() + (size_t)kWsJobMax * 64, 0.0f);
  a1.ws_r1_off = (int)plan.blob.size();
  plan.blob.resize(plan.blob.size() + 64, 0.0f);
  a1.ws_jobs = NJ;
  a1.ws_n_xt = n_xt;
  // LDS layout behind the history buffers
  const int lds_consts = kWsConstsOff;
  const int lds_tiles = lds_consts + NJ * 64;
  const int lds_xt = lds_tiles + 2 * kWsTileFloats;
  const int lds_cond = lds_xt + n_xt * 256;
  a1.ws_lds_tiles_b = lds_tiles * 4;
  a1.ws_lds_xt_b = lds_xt * 4;
  a1.ws_lds_cond_b = lds_cond * 4;
  a1.ws_lds_bytes = (lds_cond + 2 * kBlock) * 4;
  // ---- register layouts -------------------------------------------------------------------------------
  // A compute lane (g = lane / 16) keeps 4 channel values (e = 0..3) of its frame. FULL layout: channel 4g + e.
  // HALF layout (8-channel arrays): lane groups 2, 3 duplicate groups 0, 1 with the quad rotated by two,
  //   channel(g, e) = 4 (g % 2) + (e + 2 (g / 2)) % 4,
  // so that k-step m (m = 0, 1) of an MFMA can take element m of EVERY lane and still cover all 8 input
  // channels: in_channel(g, m) = 4 (g % 2) + 2 (g / 2) + m. Half the MFMAs per matrix, no cross-lane traffic.
  // Output rows are produced directly in the consumer's layout (duplicated / rotated A-tile rows).
  enum { FULL = 0, HALF = 1 };
  auto mode_of = [](int channels) { return channels == 8 ? HALF : FULL; };
  auto out_chan = [](int mode, int g, int e) { return mode == HALF ? 4 * (g % 2) + (e + 2 * (g / 2)) % 4 : 4 * g + e; };
  auto in_chan = [](int mode, int g, int m) { return mode == HALF ? 4 * (g % 2) + 2 * (g / 2) + m : 4 * g + m; };
  auto nk_of = [](int mode) { return mode == HALF ? 2 : 4; };
  // A tile of v_mfma_f32_16x16x4_f32 for the matrix W (Co x Ci, accessed through `at(co, ci)`): the record of
  // lane (g_k, row i) holds W[out_chan(i / 4, i % 4)][in_chan(g_k, m)] in element m. tile 0..2 = conv taps,
  // 3 = layer1x1; tile -1-n = extra tile n.
  int xt_next = 0;
  auto fill_tile = [&](int job, int tile, int Co, int Ci, int mode_out, int mode_in, auto at) {
    float* base = tile < 0 ? &plan.blob[(size_t)a1.ws_xt_off + (size_t)(-1 - tile) * 256]
                           : &plan.blob[(size_t)a1.ws_tiles_off + (size_t)job * kWsTileFloats + (size_t)tile * 256];
    for (int gk = 0; gk < 4; gk++)
      for (int i = 0; i < 16; i++)
        for (int m = 0; m < nk_of(mode_in); m++)
        {
          const int co = out_chan(mode_out, i / 4, i % 4), ci = in_chan(mode_in, gk, m);
          base[(gk * 16 + i) * 4 + m] = (co < Co && ci < Ci) ? at(co, ci) : 0.0f;
        }
  };
  // per-channel constants in the lane layout: entry (g, e) = v[out_chan(g, e)]
  auto fill_const = [&](int job, int vec, int n, int mode, auto at) {
    for (int g = 0; g < 4; g++)
      for (int e = 0; e < 4; e++)
      {
        const int c = out_chan(mode, g, e);
        plan.blob[(size_t)a1.ws_consts_off + (size_t)job * 64 + vec * 16 + g * 4 + e] = c < n ? at(c) : 0.0f;
      }
  };

  // where each array's pieces sit in the weight stream (same order as build_a1 walks it)
  struct Ptrs
  {
    const float* rech;
    std::vector<const float*> layer;
    const float* head;
  };
  std::vector<Ptrs> ptrs(n_arrays);
  {
    const float* w = wn.weights.data();
    for (int ai = 0; ai < n_arrays; ai++)
    {
      const LayerArraySpec& A = wn.arrays[ai];
      const int C = A.channels, K = A.kernel_sizes[0], H = A.head_size;
      ptrs[ai].rech = w;
      w += (size_t)C * A.input_size;
      for (int l = 0; l < A.num_layers(); l++)
      {
        ptrs[ai].layer.push_back(w);
        w += (size_t)C * C * K + C + C + (size_t)C * C + C;
      }
      ptrs[ai].head = w;
      w += (size_t)H * C + (A.head_bias ? H : 0);
    }
  }
  auto xw = [](int buf) { return kMfXwOff + buf * kMfXwFloats; };
  auto tb = [](int buf, int tap) { return kMfTbOff + (buf * 2 + tap) * kMfTbFloats; };
  struct JobGeo
  {
    int C = 4, d = 0, R = 64, ring_off = 0, ring_id = 0, real = 0;
  };
  std::vector<JobGeo> geo(NJ);
  std::memset(a1.cdesc, 0, sizeof(a1.cdesc));
  std::memset(a1.vdesc, 0, sizeof(a1.vdesc));
  int ji = 0;
  for (int ai = 0; ai < n_arrays; ai++)
  {
    const LayerArraySpec& A = wn.arrays[ai];
    const A1Array& arr = a1.arr[ai];
    const int C = A.channels, K = A.kernel_sizes[0], H = A.head_size;
    const int NL = A.num_layers();
    for (int l = 0; l < NL; l++, ji++)
    {
      CDesc& D = a1.cdesc[ji];
      JobGeo& G = geo[ji];
      G.C = C;
      G.d = A.dilations[l];
      G.R = arr.ring_len[l];
      G.ring_off = arr.ring_off[l];
      G.ring_id = arr.ring_id[l];
      G.real = 1;
      const int mode = mode_of(C);
      const float* w = ptrs[ai].layer[l];
      const float* cw = w; // conv [co][ci][k]
      const float* cbias = cw + (size_t)C * C * K;
      const float* mix = cbias + C; // input mixin [co] (condition size 1)
      const float* w1 = mix + C; // layer1x1 [co][ci]
      const float* b1 = w1 + (size_t)C * C;
      for (int k = 0; k < K; k++)
        fill_tile(ji, k, C, C, mode, mode, [&](int co, int ci) { return cw[((size_t)co * C + ci) * K + k]; });
      fill_tile(ji, 3, C, C, mode, mode, [&](int co, int ci) { return w1[(size_t)co * C + ci]; });
      fill_const(ji, 0, C, mode, [&](int c) { return cbias[c]; });
      fill_const(ji, 1, C, mode, [&](int c) { return mix[c]; });
      fill_const(ji, 2, C, mode, [&](int c) { return b1[c]; });
      D.flags = CD_LAYER | (mode == HALF ? CD_HALF : 0);
      D.act = A.activations[0].type;
      int g16max = 16 * (C / 4 - 1), pubmax = g16max;
      D.xt_b = a1.ws_lds_xt_b;
      // extra tile = head rechannel of src_array (+ its bias as the extra consts), outputs in layout `mode_out`
      auto head_into = [&](int src_array, int mode_out) {
        const LayerArraySpec& S = wn.arrays[src_array];
        const float* hw = ptrs[src_array].head;
        const float* hb = hw + (size_t)S.head_size * S.channels;
        const int xt = xt_next++;
        D.xt_b = a1.ws_lds_xt_b + xt * 1024;
        fill_tile(ji, -1 - xt, S.head_size, S.channels, mode_out, mode_of(S.channels),
                  [&](int h, int c) { return hw[(size_t)h * S.channels + c]; });
        fill_const(ji, 3, S.head_size, mode_out, [&](int h) { return S.head_bias ? hb[h] : 0.0f; });
      };
      if (l == 0 && ai == 0)
      {
        D.flags |= CD_X0;
        fill_const(ji, 3, C, mode, [&](int c) { return ptrs[0].rech[c]; });
        for (int co = 0; co < C; co++)
          plan.blob[(size_t)a1.ws_r1_off + co] = ptrs[0].rech[co]; // movers: natural order
      }
      else if (l == 0)
      {
        D.flags |= CD_PRE_HEAD | (mode_of(wn.arrays[ai - 1].channels) == HALF ? CD_PREV_HALF : 0);
        head_into(ai - 1, mode);
      }
      if (l == NL - 1 && ai + 1 < n_arrays)
      {
        D.flags |= CD_POST_RECH;
        const LayerArraySpec& N = wn.arrays[ai + 1];
        const float* rw = ptrs[ai + 1].rech; // [co][ci], no bias
        const int xt = xt_next++;
        D.xt_b = a1.ws_lds_xt_b + xt * 1024;
        fill_tile(ji, -1 - xt, N.channels, N.input_size, mode_of(N.channels), mode,
                  [&](int co, int ci) { return rw[(size_t)co * N.input_size + ci]; });
        pubmax = 16 * (N.channels / 4 - 1);
      }
      else if (l == NL - 1)
      {
        D.flags |= CD_POST_OUT;
        head_into(ai, FULL);
      }
      D.gp = g16max | (pubmax << 8);
      (void)H;
    }
  }
  for (int j = 0; j < NJ; j++)
  {
    const JobGeo& G = geo[j];
    const int buf = j & 1;
    CDesc& D = a1.cdesc[j];
    D.consts_b = (kWsConstsOff + j * 64) * 4;
    if (!G.real)
      D.xt_b = a1.ws_lds_xt_b;
    for (int k = 0; k < 2; k++)
    {
      const int L = G.real ? (2 - k) * G.d : 0;
      const int off = (L <= kBlock) ? xw(buf) + (kBlock - L) * kMfSC : tb(buf, k);
      (k == 0 ? D.tap0_b : D.tap1_b) = off * 4;
    }
    D.pub_b = (xw(buf ^ 1) + kBlock * kMfSC) * 4;
    VDesc& V = a1.vdesc[j];
    const int nbuf = (j + 1) & 1;
    const JobGeo& N = geo[(j + 1) % NJ]; // successor: its history is dropped during job j
    const JobGeo& F = geo[(j + 1 + PF) % NJ];
    // which sets a job needs (plan.h, VDesc)
    auto sets = [&](const JobGeo& G, int& LA, int& LB, int& dst_a, int& dst_b, int buf) {
      LA = kBlock, LB = 0, dst_a = xw(buf), dst_b = tb(buf, 0);
      if (!G.real || 2 * G.d <= kBlock)
        return;
      if (G.d <= kBlock)
        LB = 2 * G.d;
      else
      {
        LA = 2 * G.d, LB = G.d;
        dst_a = tb(buf, 0), dst_b = tb(buf, 1);
      }
    };
    int LA, LB, da, db;
    sets(N, LA, LB, da, db, nbuf);
    V.flags = (G.real ? MV_RING : 0) | (j == NJ - 1 ? MV_SUCC_FIRST : 0) | (LB ? MV_SUCC_B : 0);
    V.st_a_b = da * 4;
    V.st_b_b = db * 4;
    V.st_x0_b = (xw(nbuf) + kBlock * kMfSC) * 4;
    sets(F, LA, LB, da, db, 0);
    V.f_rbase = F.real ? F.ring_off * 4 : 0;
    V.f_R = F.real ? F.R : 64;
    V.f_LA = F.real ? LA : 64;
    V.f_LB = F.real ? LB : 0;
    V.f_ring_id = F.real ? F.ring_id : 0;
    V.f_q16max = F.real ? 16 * (F.C / 4 - 1) : 0;
    V.ap_src_b = (xw(buf) + kBlock * kMfSC) * 4;
    V.ring_b = G.ring_off * 4;
    V.R = G.real ? G.R : 64;
    V.ring_id = G.real ? G.ring_id : 0;
    V.q16max = 16 * (G.C / 4 - 1);
  }
  a1.ws_ok = 1;
}

// Job table of the interleaved-frame mapping (plan.h: IlDesc / IlFetch; nam_a1_p2_kernel's compile-time tables are checked against it). Built from the finished A1 plan: same eligibility, tiles,
// constants and per-job flags as nam_a1_mfma_kernel; ring offsets are final (write-position table included).
void build_a1_il(Plan& plan)
{
  A1Plan& a1 = plan.a1;
  a1.il_ok = 0;
  if (!a1.valid || !a1.ws_ok)
    return;
  int n_layers = 0;
  for (int ai = 0; ai < a1.n_arrays; ai++)
    n_layers += a1.arr[ai].n_layers;
  // the kernel's job loop is unrolled 10 deep: jobs per block are padded to a multiple of 10 with idle jobs
  const int D = 10;
  const int NJ = (n_layers + 9) / 10 * 10;
  if (NJ > kIlJobMax || n_layers > kWsJobMax)
    return;
  struct Geo
  {
    int C = 4, d = 1, R = 64, ring_b = 0, ring_id = 0, kind = IL_IDLE;
  };
  std::vector<Geo> geo((size_t)NJ);
  int j = 0, n_exch = 0;
  for (int ai = 0; ai < a1.n_arrays; ai++)
    for (int l = 0; l < a1.arr[ai].n_layers; l++, j++)
    {
      Geo& G = geo[(size_t)j];
      const A1Array& A = a1.arr[ai];
      G.C = A.channels;
      G.d = A.dil[l];
      G.R = A.ring_len[l];
      G.ring_b = A.ring_off[l] * 4;
      G.ring_id = A.ring_id[l];
      if (G.d >= kBlock)
      {
        // every tap lies in an earlier block. Across a block boundary of one launch a lane may only re-read rows it
        // stored itself (same-wave program order is the only ordering there is without a barrier): lookbacks must be
        // whole blocks
        if (G.d % kBlock != 0)
          return;
        G.kind = IL_HIST;
      }
      else if (G.d == 4 || G.d == 8 || G.d == 16 || G.d == 32)
        G.kind = IL_DPP;
      else
      {
        if (2 * G.d > kBlock && (2 * G.d) % kBlock != 0)
          return; // tap 0 would come from the ring at a lookback that is not a whole number of blocks (see above)
        G.kind = IL_EXCH;
        n_exch++;
      }
    }
  a1.il_jobs = NJ;
  a1.il_real_jobs = n_layers;
  a1.il_depth = D;
  a1.il_exch = n_exch;
  // LDS: exchange windows [2] | constants [jobs][64] | extra tiles [n][256] | tiles [jobs][1024] | loader progress word
  a1.il_consts_b = 2 * kIlWinB;
  a1.il_xt_b = a1.il_consts_b + n_layers * 256;
  a1.il_tiles_b = a1.il_xt_b + a1.ws_n_xt * 1024;
  a1.il_flag_b = a1.il_tiles_b + n_layers * kWsTileFloats * 4;
  a1.il_lds_bytes = a1.il_flag_b + 64;
  if (a1.il_lds_bytes > 160 * 1024)
    return;
  std::memset(a1.il_desc, 0, sizeof(a1.il_desc));
  std::memset(a1.il_fetch, 0, sizeof(a1.il_fetch));
  auto xt_of = [&](int job) { return a1.il_xt_b + (a1.cdesc[job].xt_b - a1.ws_lds_xt_b); };
  for (j = 0; j < NJ; j++)
  {
    const Geo& G = geo[(size_t)j];
    IlDesc& Dd = a1.il_desc[j];
    Dd.kind = G.kind;
    if (G.kind != IL_IDLE)
    {
      Dd.flags = a1.cdesc[j].flags;
      Dd.act = a1.cdesc[j].act;
      Dd.gp = a1.cdesc[j].gp & 0xff;
      Dd.ring_b = G.ring_b;
      Dd.R = G.R;
      Dd.ring_id = G.ring_id;
      Dd.row_b = G.C * 4;
      Dd.dil = G.d;
      Dd.tap0_lds = (G.kind == IL_EXCH && 2 * G.d <= kBlock) ? 1 : 0;
    }
    else
    {
      Dd.R = kBlock;
      Dd.row_b = 16;
    }
    // operands of the next real job (padding jobs pass job 0's along: they sit at the end of the block)
    const int nj = (j + 1) % NJ;
    const int nreal = geo[(size_t)nj].kind != IL_IDLE ? nj : 0;
    Dd.n_consts_b = a1.il_consts_b + nreal * 256;
    Dd.n_xt_b = xt_of(nreal);
    Dd.n_tiles_b = a1.il_tiles_b + nreal * kWsTileFloats * 4;
    Dd.n_ready = nreal + 1;
    // requests for the job D ahead
    const Geo& F = geo[(size_t)((j + D) % NJ)];
    IlFetch& Ff = a1.il_fetch[j];
    Ff.ring_b = F.ring_b;
    Ff.R = F.kind != IL_IDLE ? F.R : kBlock;
    Ff.ring_id = F.kind != IL_IDLE ? F.ring_id : 0;
    Ff.row_b = F.kind != IL_IDLE ? F.C * 4 : 16;
    Ff.nA = Ff.nB = 16;
    switch (F.kind)
    {
      case IL_HIST:
        Ff.LA = 2 * F.d;
        Ff.LB = F.d;
        break;
      case IL_DPP:
        Ff.LA = 2 * F.d;
        Ff.LB = F.d;
        Ff.nA = std::min(16, F.d / 2);
        Ff.nB = F.d / 4;
        break;
      case IL_EXCH:
        Ff.LA = kBlock; // the lane's own frame of the previous block, for the "previous" half of the LDS window
        Ff.LB = 2 * F.d > kBlock ? 2 * F.d : 0; // tap 0 from the ring
        break;
      default: Ff.LA = Ff.LB = 0; break;
    }
  }
  a1.il_ok = 1;
  // the official topology with compile-time tables (plan.h: namespace p2): only if those tables ARE this model's
  a1.p2_ok = 0;
  if (a1.n_arrays == 2 && n_layers == p2::kJobs && NJ == p2::kJobs && D == p2::kDepth && a1.ws_n_xt == p2::kXt
      && a1.il_consts_b == p2::kConstsB && a1.il_xt_b == p2::kXtB && a1.il_tiles_b == p2::kTilesB
      && a1.il_flag_b == p2::kFlagB && a1.arr[0].act == a1.arr[1].act)
  {
    const int C0 = a1.arr[0].channels, C1 = a1.arr[1].channels;
    bool same = (C0 == 16 && C1 == 8) || (C0 == 12 && C1 == 8) || (C0 == 8 && C1 == 4); // instantiated in kernel_a1_p2.hip
    for (j = 0; same && j < NJ; j++)
    {
      const IlDesc e = p2::desc(C0, C1, a1.arr[0].act, j);
      const IlFetch f = p2::fetch(C0, C1, j);
      same = std::memcmp(&e, &a1.il_desc[j], sizeof(e)) == 0 && std::memcmp(&f, &a1.il_fetch[j], sizeof(f)) == 0;
    }
    if (same)
    {
      a1.p2_ok = 1;
      a1.p2_c0 = C0;
      a1.p2_c1 = C1;
    }
  }
}

void build_a1(const WaveNetSpec& wn, Plan& plan)
{
  A1Plan& a1 = plan.a1;
  a1.valid = 0;
  if (wn.condition_dsp || wn.with_head || wn.in_channels != 1)
    return;
  if (wn.arrays.empty() || (int)wn.arrays.size() > kA1MaxArrays)
    return;
  if (wn.arrays.back().head_size != 1)
    return;
  for (size_t ai = 0; ai < wn.arrays.size(); ai++)
  {
    const LayerArraySpec& A = wn.arrays[ai];
    if (A.condition_size != 1 || A.groups_input != 1 || A.groups_input_mixin != 1 || !A.layer1x1_active
        || A.layer1x1_groups != 1 || A.head1x1_active || A.bottleneck != A.channels)
      return;
    // a head rechannel with taps (A2: K = 16) is handled for a single output channel
    if (A.head_kernel_size < 1 || A.head_kernel_size > 16 || (A.head_kernel_size > 1 && A.head_size != 1))
      return;
    if (!a1_channel_supported(A.channels) || A.num_layers() < 1 || A.num_layers() > kA1MaxLayers)
      return;
    if (ai + 1 < wn.arrays.size() && !a1_channel_supported(A.head_size))
      return;
    for (int k = 0; k < FILM_COUNT; k++)
      if (A.film[k].active)
        return;
    for (int l = 0; l < A.num_layers(); l++)
    {
      if (A.gating_modes[l] != GATING_NONE || A.kernel_sizes[l] < 1 || A.kernel_sizes[l] > 16)
        return;
      const ActSpec& a = A.activations[l];
      const ActSpec& a0 = A.activations[0];
      if (a.type != a0.type || a.type == ACT_PRELU || a.type == ACT_LEAKYHARDTANH || a.type == ACT_LUT || a.p[0] != a0.p[0])
        return;
    }
  }
  // The fast kernel shares the generic plan's state layout: ring r of the generic program is the
  // r-th dilated conv in execution order, i.e. (array, layer) order here (head rechannel has K = 1).
  const float* w = wn.weights.data();
  int ring_id = 0;
  int state_off = 0;
  for (size_t ai = 0; ai < wn.arrays.size(); ai++)
  {
    const LayerArraySpec& A = wn.arrays[ai];
    A1Array& out = a1.arr[ai];
    std::memset(&out, 0, sizeof(out));
    const int C = A.channels, H = A.head_size, KH = A.head_kernel_size;
    out.in_size = A.input_size;
    out.channels = C;
    out.kernel = A.kernel_sizes[0];
    out.n_layers = A.num_layers();
    out.head_size = H;
    out.act = A.activations[0].type;
    out.act_p0 = A.activations[0].p[0];
    size_t total = (size_t)A.input_size * C;
    for (int l = 0; l < out.n_layers; l++)
      total += (size_t)A.kernel_sizes[l] * C * C + C + C + (size_t)C * C + C;
    out.layer_stride = 0; // per-layer kernel sizes: see layer_off
    total += (size_t)KH * C * H + H + 1;
    while (plan.blob.size() % 16)
      plan.blob.push_back(0.0f);
    out.w_base = (int)plan.blob.size();
    plan.blob.resize(plan.blob.size() + total + 16, 0.0f);
    float* const base = plan.blob.data() + out.w_base;
    float* dst = base;
    // rechannel: stream [co][ci] -> packed [ci][co]
    for (int co = 0; co < C; co++)
      for (int ci = 0; ci < A.input_size; ci++)
        dst[(size_t)ci * C + co] = *(w++);
    dst += (size_t)A.input_size * C;
    auto add_ring = [&](int lookback, int& off, int& len, int& id) {
      if (lookback > 0)
      {
        len = lookback + kBlock;
        off = state_off;
        id = ring_id;
        if (ring_id < 64)
          a1.ring_len_by_id[ring_id] = len;
        ring_id++;
        state_off += C * len;
      }
      else
      {
        len = 0;
        off = 0;
        id = -1;
      }
    };
    for (int l = 0; l < out.n_layers; l++)
    {
      const int K = A.kernel_sizes[l];
      out.ksize[l] = K;
      out.layer_off[l] = (int)(dst - base);
      float* cw = dst;
      for (int co = 0; co < C; co++)
        for (int ci = 0; ci < C; ci++)
          for (int k = 0; k < K; k++)
            cw[((size_t)k * C + ci) * C + co] = *(w++);
      float* cb = cw + (size_t)K * C * C;
      for (int co = 0; co < C; co++)
        cb[co] = *(w++);
      float* mx = cb + C;
      for (int co = 0; co < C; co++)
        mx[co] = *(w++);
      float* w1 = mx + C;
      for (int co = 0; co < C; co++)
        for (int ci = 0; ci < C; ci++)
          w1[(size_t)ci * C + co] = *(w++);
      float* b1 = w1 + (size_t)C * C;
      for (int co = 0; co < C; co++)
        b1[co] = *(w++);
      dst = b1 + C;
      out.dil[l] = A.dilations[l];
      add_ring((K - 1) * A.dilations[l], out.ring_off[l], out.ring_len[l], out.ring_id[l]);
    }
    // head rechannel (a Conv1D, model.cpp:399-400): stream [h][c][k] (+ bias[h]) -> packed [k][c][h], bias[h]
    out.head_k = KH;
    out.head_dil = A.head_dilation;
    out.head_off = (int)(dst - base);
    for (int h = 0; h < H; h++)
      for (int c = 0; c < C; c++)
        for (int k = 0; k < KH; k++)
          dst[((size_t)k * C + c) * H + h] = *(w++);
    float* hb = dst + (size_t)KH * C * H;
    for (int h = 0; h < H; h++)
      hb[h] = A.head_bias ? *(w++) : 0.0f;
    add_ring((KH - 1) * A.head_dilation, out.head_ring_off, out.head_ring_len, out.head_ring_id);
  }
  a1.n_arrays = (int)wn.arrays.size();
  a1.n_rings = ring_id;
  a1.head_scale_off = (int)plan.blob.size();
  plan.blob.push_back(*(w++));
  if (w != wn.weights.data() + wn.weights.size() || ring_id != plan.n_rings || ring_id > 64)
  {
    a1.valid = 0; // layouts disagree: keep the generic path only
    return;
  }
  a1.valid = 1;

  build_a1_ws(wn, plan);
}

// Chunk table and MFMA operand tiles of nam_kt_mfma_kernel (plan.h: KtDesc), derived from the packed A1 weights and
// ring geometry that build_a1 has already laid down (ring offsets final, i.e. behind the write-position table).
// Requires a single layer array with channels % 4 == 0 (<= 16), a mono input and a single head output channel; any
// per-layer kernel size and head kernel size up to 16.
void build_a1_kt(Plan& plan)
{
  A1Plan& a1 = plan.a1;
  a1.kt_ok = 0;
  if (!a1.valid || a1.n_arrays != 1)
    return;
  const A1Array A = a1.arr[0]; // by value: the blob grows below
  const int C = A.channels, NL = A.n_layers;
  if (A.in_size != 1 || C % 4 != 0 || C > 16 || A.head_size != 1)
    return;
  enum { FULL = 0, HALF = 1 };
  const int mode = C == 8 ? HALF : FULL;
  const int NK = mode == HALF ? 2 : 4;
  auto out_chan = [&](int g, int e) { return mode == HALF ? 4 * (g % 2) + (e + 2 * (g / 2)) % 4 : 4 * g + e; };
  auto in_chan = [&](int g, int m) { return mode == HALF ? 4 * (g % 2) + 2 * (g / 2) + m : 4 * g + m; };
  auto chunks_of = [](int K) { return (K + kKtTaps - 1) / kKtTaps; };
  int n_chunks = chunks_of(A.head_k);
  for (int l = 0; l < NL; l++)
    n_chunks += chunks_of(A.ksize[l]);
  // the kernel requests operands up to 5 chunks ahead; a chunk of the NEXT block must lie well behind the current one
  // so that the rows it reads from THIS block are long written
  if (n_chunks > kKtChunkMax || n_chunks < 16)
    return;
  // blob regions: tap tiles [chunk][kKtTaps taps][64 lanes][NK] (half layout: taps paired per lane, see below),
  // then what the kernel keeps in LDS: 1x1 tiles [layer][64 lanes][NK] | constants [layer + head][3][16]
  const int tile_floats = 64 * NK;
  const int chunk_floats = kKtTaps * tile_floats;
  while (plan.blob.size() % 64)
    plan.blob.push_back(0.0f);
  const size_t tiles0 = plan.blob.size();
  plan.blob.resize(tiles0 + (size_t)(n_chunks + 1) * chunk_floats, 0.0f);
  const size_t w1_0 = plan.blob.size();
  plan.blob.resize(w1_0 + (size_t)NL * tile_floats, 0.0f);
  const size_t consts0 = plan.blob.size();
  plan.blob.resize(consts0 + (size_t)(NL + 1) * 48, 0.0f);
  const size_t lds_end = plan.blob.size();
  const size_t rech0 = lds_end;
  plan.blob.resize(rech0 + 16, 0.0f);
  float* const blob = plan.blob.data();
  const float* const base = blob + A.w_base;
  // value m of lane (gk, i) of a tile: W[out_chan(i / 4, i % 4)][in_chan(gk, m)]
  auto tile_value = [&](int lane, int m, auto at) {
    const int gk = lane / 16, i = lane % 16;
    const int co = out_chan(i / 4, i % 4), ci = in_chan(gk, m);
    return (co < C && ci < C) ? at(co, ci) : 0.0f;
  };
  // tap `slot` of a chunk record. Full layout: [slot][lane][4 k-steps]. Half layout (2 k-steps): the taps are
  // paired so that one 16-byte load per lane brings two taps: [slot / 2][lane][(slot % 2) * 2 + m].
  auto fill_tap = [&](size_t chunk_off, int slot, auto at) {
    for (int lane = 0; lane < 64; lane++)
      for (int m = 0; m < NK; m++)
      {
        const size_t idx = NK == 2 ? (size_t)(slot / 2) * 256 + (size_t)lane * 4 + (slot % 2) * 2 + m
                                   : (size_t)slot * 256 + (size_t)lane * 4 + m;
        blob[chunk_off + idx] = tile_value(lane, m, at);
      }
  };
  auto fill_const = [&](size_t off, auto at) {
    for (int g = 0; g < 4; g++)
      for (int e = 0; e < 4; e++)
      {
        const int c = out_chan(g, e);
        blob[off + (size_t)g * 4 + e] = c < C ? at(c) : 0.0f;
      }
  };
  fill_const(rech0, [&](int c) { return base[c]; }); // rechannel [ci = 0][co]
  int chunk = 0;
  // tap(k)(co, ci): weight of tap k
  auto emit_layer = [&](int K, int d, int flags, int w1_lds_b, int consts_lds_b, int ring_off, int R, int ring_id,
                        auto tap) {
    for (int c0 = 0; c0 < K; c0 += kKtTaps)
    {
      const size_t chunk_off = tiles0 + (size_t)chunk * chunk_floats;
      KtDesc& D = a1.kt_desc[chunk++];
      std::memset(&D, 0, sizeof(D));
      D.ntaps = std::min(kKtTaps, K - c0);
      D.flags = flags | (c0 == 0 ? (int)KT_FIRST | (ring_id >= 0 ? (int)KT_RING : 0) : 0)
                | (c0 + kKtTaps >= K ? (int)KT_LAST : 0);
      if (!(D.flags & KT_LAST))
        D.flags &= ~(int)KT_NEXT_HEAD;
      D.tile_off = (int)chunk_off;
      D.w1_off = w1_lds_b;
      D.consts_off = consts_lds_b;
      D.ring_b = ring_id >= 0 ? ring_off * 4 : 0;
      D.R = ring_id >= 0 ? R : kKtNoTap;
      D.ring_id = ring_id >= 0 ? ring_id : 0;
      for (int i = 0; i < kKtTaps; i++)
      {
        D.L[i] = i < D.ntaps ? (K - 1 - (c0 + i)) * d : kKtNoTap;
        if (i < D.ntaps)
        {
          const int k = c0 + i;
          fill_tap(chunk_off, i, [&](int co, int ci) { return tap(k, co, ci); });
        }
      }
    }
  };
  for (int l = 0; l < NL; l++)
  {
    const int K = A.ksize[l];
    const float* cw = base + A.layer_off[l];
    const float* cb = cw + (size_t)K * C * C;
    const float* mx = cb + C;
    const float* w1 = mx + C;
    const float* b1 = w1 + (size_t)C * C;
    const size_t w1_off = w1_0 + (size_t)l * tile_floats;
    for (int lane = 0; lane < 64; lane++)
      for (int m = 0; m < NK; m++)
        blob[w1_off + (size_t)lane * NK + m] = tile_value(lane, m, [&](int co, int ci) { return w1[(size_t)ci * C + co]; });
    const size_t co_off = consts0 + (size_t)l * 48;
    fill_const(co_off, [&](int c) { return cb[c]; });
    fill_const(co_off + 16, [&](int c) { return mx[c]; });
    fill_const(co_off + 32, [&](int c) { return b1[c]; });
    emit_layer(K, A.dil[l], l + 1 == NL ? (int)KT_NEXT_HEAD : 0, (int)((w1_off - w1_0) * 4), (int)((co_off - w1_0) * 4),
               A.ring_off[l], A.ring_len[l], A.ring_id[l],
               [&](int k, int co, int ci) { return cw[((size_t)k * C + ci) * C + co]; });
  }
  {
    // head rechannel: [k][c][h = 0] -> output row 0 only; bias rides in the "conv bias" slot
    const int K = A.head_k;
    const float* hw = base + A.head_off;
    const float* hb = hw + (size_t)K * C;
    const size_t co_off = consts0 + (size_t)NL * 48;
    fill_const(co_off, [&](int c) { return c == 0 ? hb[0] : 0.0f; });
    emit_layer(K, A.head_dil, (int)KT_HEAD, 0, (int)((co_off - w1_0) * 4), A.head_ring_off, A.head_ring_len,
               A.head_ring_id, [&](int k, int co, int ci) { return co == 0 ? hw[(size_t)k * C + ci] : 0.0f; });
  }
  a1.kt_chunks = chunk;
  a1.kt_nk = NK;
  a1.kt_rech_off = (int)rech0;
  a1.kt_lds_src_off = (int)w1_0;
  a1.kt_lds_floats = (int)(lds_end - w1_0);
  a1.kt_ok = 1;
}

// nam_kq_kernel (kernel_kq.hip) is compiled for ONE topology (kp_table.h: by default the A2 stack the reference's fused
// path is written for, a2_fast.cpp:57-764): it may run a model only when the K-tap kernel's plan of that model is, layer by
// layer, what the kernel's compile-time tables say — kernel sizes, dilations, ring geometry and offsets, chunk and tile
// offsets, the LDS block.
void build_a1_kp(Plan& plan)
{
  A1Plan& a1 = plan.a1;
  a1.kp_ok = 0;
  if (!a1.valid || !a1.kt_ok || a1.kt_nk != 2 || a1.n_arrays != 1)
    return;
  const A1Array& A = a1.arr[0];
  if (A.channels != kp::kC || A.n_layers != kp::kLayers || A.head_k != kp::kKs[kp::kLayers] || A.head_dil != kp::kDs[kp::kLayers]
      || A.head_ring_id != kp::kLayers || a1.n_rings != kp::kJobs || a1.kt_chunks != kp::kChunks
      || a1.kt_lds_floats != kp::kLayers * 128 + kp::kJobs * 48)
    return;
  for (int l = 0; l < kp::kLayers; l++)
    if (A.ksize[l] != kp::kKs[l] || A.dil[l] != kp::kDs[l] || A.ring_id[l] != l || A.ring_len[l] != kp::ring_len(l)
        || A.ring_off[l] != kp::ring_off(l))
      return;
  if (A.head_ring_len != kp::ring_len(kp::kLayers) || A.head_ring_off != kp::ring_off(kp::kLayers))
    return;
  const int tiles0 = a1.kt_desc[0].tile_off;
  for (int j = 0; j < kp::kJobs; j++)
  {
    const KtDesc& D = a1.kt_desc[kp::chunk0(j)];
    if (D.tile_off != tiles0 + kp::chunk0(j) * kKtTaps * 128 || !(D.flags & KT_FIRST) || D.ring_b != kp::ring_off(j) * 4 || D.R != kp::ring_len(j)
        || D.consts_off != kp::kLayers * 512 + j * 192 || (j < kp::kLayers && D.w1_off != j * 512))
      return;
  }
  // nam_kq_kernel's weight block (kernel_kq.hip): v_mfma_f32_4x4x1_16b operands for a lane-per-frame layout. One 256-byte
  // tile per tap, [lane class i = lane % 4][h][c] = W[out = 4 h + i][in = c]; every job's taps in order, then the layers'
  // 1x1; constants per job bias[8] | mixin[8] | 1x1 bias[8]; the rechannel column [8]. The head rechannel has one output:
  // class 0, half 0 only.
  {
    const int C = kp::kC;
    while (plan.blob.size() % 64)
      plan.blob.push_back(0.0f);
    const size_t w0 = plan.blob.size();
    int n_taps = 0;
    for (int j = 0; j < kp::kJobs; j++)
      n_taps += kp::kKs[j];
    const size_t c_0 = w0 + (size_t)(n_taps + kp::kLayers) * 64, rech0 = c_0 + (size_t)kp::kJobs * 24;
    plan.blob.resize(rech0 + 16, 0.0f);
    float* const blob = plan.blob.data();
    const A1Array& AA = a1.arr[0]; // (the vector may have moved: take the array again)
    const float* const base = blob + AA.w_base;
    auto fill_tile = [&](size_t off, auto at) { // at(co, ci)
      for (int i = 0; i < 4; i++)
        for (int h = 0; h < 2; h++)
          for (int c = 0; c < C; c++)
            blob[off + (size_t)i * 16 + h * 8 + c] = at(4 * h + i, c);
    };
    size_t tile = w0;
    const size_t w1_0 = w0 + (size_t)n_taps * 64;
    for (int l = 0; l < kp::kLayers; l++)
    {
      const int K = AA.ksize[l];
      const float* cw = base + AA.layer_off[l];
      const float* cb = cw + (size_t)K * C * C;
      const float* mx = cb + C;
      const float* w1 = mx + C;
      const float* b1 = w1 + (size_t)C * C;
      for (int k = 0; k < K; k++, tile += 64)
        fill_tile(tile, [&](int co, int ci) { return cw[((size_t)k * C + ci) * C + co]; });
      fill_tile(w1_0 + (size_t)l * 64, [&](int co, int ci) { return w1[(size_t)ci * C + co]; });
      for (int c = 0; c < C; c++)
      {
        blob[c_0 + (size_t)l * 24 + c] = cb[c];
        blob[c_0 + (size_t)l * 24 + 8 + c] = mx[c];
        blob[c_0 + (size_t)l * 24 + 16 + c] = b1[c];
      }
    }
    {
      const int K = AA.head_k;
      const float* hw = base + AA.head_off;
      const float* hb = hw + (size_t)K * C;
      for (int k = 0; k < K; k++, tile += 64)
        fill_tile(tile, [&](int co, int ci) { return co == 0 ? hw[(size_t)k * C + ci] : 0.0f; });
      blob[c_0 + (size_t)kp::kLayers * 24] = hb[0];
    }
    for (int c = 0; c < C; c++)
      blob[rech0 + c] = base[c]; // rechannel [ci = 0][co]
    a1.kq_w_off = (int)w0;
  }
  a1.kp_ok = 1;
}

// nam_a1_q_kernel (kernel_a1_q.hip) runs this model if it IS the topology of aq_table.h: the official two-array stack with
// 16 and 8 channels. Its weight block (aq_table.h: kWrOff .. kBlockFloats) holds the lane-per-frame (4x4x1) tiles of the
// transition, of array 1 and of the head, every constant, and array 0's constants in channel order; array 0's matrices
// are the FULL-layout tiles build_a1_ws has already packed (ws_tiles_off), which the kernel keeps in registers.
void build_a1_q(Plan& plan)
{
  A1Plan& a1 = plan.a1;
  a1.q_ok = 0;
  a1.q_w_off = 0;
  if (!a1.valid || !a1.p2_ok || a1.p2_c0 != aq::kC0 || a1.p2_c1 != aq::kC1 || a1.n_arrays != 2 || a1.n_rings != aq::kRings)
    return;
  for (int ai = 0; ai < 2; ai++)
  {
    const A1Array& A = a1.arr[ai];
    if (A.channels != (ai == 0 ? aq::kC0 : aq::kC1) || A.n_layers != aq::kLayers || A.head_k != 1
        || A.in_size != (ai == 0 ? 1 : aq::kC0) || A.head_size != (ai == 0 ? aq::kC1 : 1))
      return;
    for (int l = 0; l < aq::kLayers; l++)
    {
      const int job = ai == 0 ? l : aq::kJobM0 + l;
      if (A.ksize[l] != 3 || A.dil[l] != aq::dil(job) || A.ring_id[l] != aq::ring_id(job) || A.ring_len[l] != aq::ring_len(job)
          || A.ring_off[l] != aq::ring_off(job))
        return;
    }
  }
  while (plan.blob.size() % 64)
    plan.blob.push_back(0.0f);
  const size_t w0 = plan.blob.size();
  plan.blob.resize(w0 + (size_t)aq::kBlockFloats, 0.0f);
  float* const q = plan.blob.data() + w0;
  const A1Array& A0 = a1.arr[0];
  const A1Array& A1 = a1.arr[1];
  const float* const base0 = plan.blob.data() + A0.w_base;
  const float* const base1 = plan.blob.data() + A1.w_base;
  // 4x4x1 tile: [lane class i][h][c] = W[out = 4 h + i][in = c]
  auto fill_tile = [&](float* t, int n_half, int n_in, auto at) {
    for (int i = 0; i < 4; i++)
      for (int h = 0; h < n_half; h++)
        for (int c = 0; c < n_in; c++)
          t[(i * n_half + h) * n_in + c] = at(4 * h + i, c);
  };
  const int C0 = aq::kC0, C1 = aq::kC1;
  // array 1's rechannel: packed [ci][co]; array 0's head rechannel: packed [k = 0][c][h], bias[h] behind it
  fill_tile(q + aq::kWrOff, 2, C0, [&](int co, int ci) { return base1[(size_t)ci * C1 + co]; });
  const float* hw0 = base0 + A0.head_off;
  fill_tile(q + aq::kWhOff, 2, C0, [&](int co, int ci) { return hw0[(size_t)ci * C1 + co]; });
  for (int h = 0; h < C1; h++)
    q[aq::kTConsts + h] = hw0[(size_t)C0 * C1 + h];
  for (int l = 0; l < aq::kLayers; l++)
  {
    const float* cw = base1 + A1.layer_off[l];
    const float* cb = cw + (size_t)3 * C1 * C1;
    const float* mx = cb + C1;
    const float* w1 = mx + C1;
    const float* b1 = w1 + (size_t)C1 * C1;
    for (int k = 0; k < 3; k++)
      fill_tile(q + aq::kMTiles + (l * 4 + k) * aq::kTileM, 2, C1, [&](int co, int ci) { return cw[((size_t)k * C1 + ci) * C1 + co]; });
    fill_tile(q + aq::kMTiles + (l * 4 + 3) * aq::kTileM, 2, C1, [&](int co, int ci) { return w1[(size_t)ci * C1 + co]; });
    for (int c = 0; c < C1; c++)
    {
      q[aq::kMConsts + l * 24 + c] = cb[c];
      q[aq::kMConsts + l * 24 + 8 + c] = mx[c];
      q[aq::kMConsts + l * 24 + 16 + c] = b1[c];
    }
  }
  {
    const float* hw1 = base1 + A1.head_off; // [k = 0][c][h = 0], bias behind it
    fill_tile(q + aq::kHeadTile, 2, C1, [&](int co, int ci) { return co == 0 ? hw1[ci] : 0.0f; });
    q[aq::kTConsts + 8] = hw1[C1];
  }
  for (int l = 0; l < aq::kLayers; l++)
  {
    const float* cw = base0 + A0.layer_off[l];
    const float* cb = cw + (size_t)3 * C0 * C0;
    const float* mx = cb + C0;
    const float* w1 = mx + C0;
    const float* b1 = w1 + (size_t)C0 * C0;
    float* d = q + aq::kBigConsts + l * 64;
    for (int c = 0; c < C0; c++)
    {
      d[c] = cb[c];
      d[16 + c] = mx[c];
      d[32 + c] = b1[c];
      d[48 + c] = l == 0 ? base0[c] : 0.0f; // array 0's rechannel column [ci = 0][co]
    }
  }
  a1.q_w_off = (int)w0;
  a1.q_ok = 1;
}

// The official "lite" size (12 -> 6 channels) misses the matrix-core kernel only because 6 is not a multiple of 4.
// Zero-padding such arrays to the next multiple (weights, biases, mixin, rechannels all zero for the extra channels)
// is exact on the real channels: the padded ones carry f(0) through the activations and meet zero weights everywhere.
// Returns false when the model is not a plain kernel-size-3 WaveNet that padding would help.
//
// The official topology — two arrays of ten layers, kernel size 3, dilations 1 .. 512, Tanh — at the smaller official widths
// (lite 12 / 6, feather 8 / 4; NAM's "standard" is 16 / 8) is padded all the way to 16 / 8: nam_a1_q_kernel (kernel_a1_q.hip,
// compiled for that one shape) then takes it, and its 6.8 us per buffer at 256 streams beats what the narrower shapes reach on
// nam_a1_p4_kernel (lite 8.2, feather 7.1: profiles/r05/official_sizes_256.txt) — the matrix pipe does not care about rows of
// zeros as much as the pipeline cares about LDS-resident rings. (nano, 4 / 2, stays on nam_wn_reg_kernel: 5.7 us.)
bool official_standard_topology(const WaveNetSpec& wn)
{
  if (wn.arrays.size() != 2)
    return false;
  for (const LayerArraySpec& A : wn.arrays)
  {
    if (A.num_layers() != 10)
      return false;
    for (int l = 0; l < 10; l++)
      if (A.dilations[(size_t)l] != (1 << l) || (A.activations[(size_t)l].type != ACT_TANH && A.activations[(size_t)l].type != ACT_FASTTANH)
          || A.activations[(size_t)l].type != wn.arrays[0].activations[0].type)
        return false;
  }
  const int c0 = wn.arrays[0].channels, c1 = wn.arrays[1].channels;
  return c0 >= 8 && c0 <= 16 && c1 >= 4 && c1 <= 8 && !(c0 == 16 && c1 == 8);
}
bool pad_channels_for_mfma(const WaveNetSpec& wn, WaveNetSpec& out)
{
  if (wn.condition_dsp || wn.with_head || wn.in_channels != 1 || wn.slimmable || wn.arrays.empty())
    return false;
  const bool to_standard = official_standard_topology(wn);
  auto padw = [&](size_t array, int c) { return to_standard ? (array == 0 ? 16 : 8) : (c + 3) / 4 * 4; }; // padded width of an array
  bool any = to_standard;
  for (const LayerArraySpec& A : wn.arrays)
  {
    // (1-3 channels stay as they are: for models that small the VALU kernel is the better one once the chip is full)
    if ((A.channels % 4 != 0 && A.channels < 5) || A.channels > 16 || A.bottleneck != A.channels || A.condition_size != 1
        || A.groups_input != 1 || A.groups_input_mixin != 1 || !A.layer1x1_active || A.layer1x1_groups != 1
        || A.head1x1_active || A.head_kernel_size != 1)
      return false;
    for (int k : A.kernel_sizes)
      if (k != 3)
        return false;
    for (int g : A.gating_modes)
      if (g != GATING_NONE)
        return false;
    for (int k = 0; k < FILM_COUNT; k++)
      if (A.film[k].active)
        return false;
    any = any || A.channels % 4 != 0;
  }
  if (!any)
    return false;
  out = wn;
  out.weights.clear();
  const float* w = wn.weights.data();
  const size_t n_arr = wn.arrays.size();
  for (size_t ai = 0; ai < n_arr; ai++)
  {
    const LayerArraySpec& A = wn.arrays[ai];
    LayerArraySpec& P = out.arrays[ai];
    const int C = A.channels, Cp = padw(ai, C);
    const int in = A.input_size, inp = ai == 0 ? in : padw(ai - 1, wn.arrays[ai - 1].channels);
    const int H = A.head_size, Hp = ai + 1 < n_arr ? padw(ai + 1, wn.arrays[ai + 1].channels) : H;
    if (ai > 0 && in != wn.arrays[ai - 1].channels)
      return false;
    if (ai + 1 < n_arr && H != wn.arrays[ai + 1].channels)
      return false;
    P.channels = P.bottleneck = Cp;
    P.input_size = inp;
    P.head_size = Hp;
    // tensor [rows][cols][taps] of the stream, zero-padded to [rows_p][cols_p][taps]
    auto tensor = [&](int rows, int cols, int taps, int rows_p, int cols_p) {
      for (int r = 0; r < rows_p; r++)
        for (int c = 0; c < cols_p; c++)
          for (int k = 0; k < taps; k++)
            out.weights.push_back(r < rows && c < cols ? w[((size_t)r * cols + c) * taps + k] : 0.0f);
      w += (size_t)rows * cols * taps;
    };
    tensor(C, in, 1, Cp, inp); // rechannel [C][in]
    for (int l = 0; l < A.num_layers(); l++)
    {
      tensor(C, C, A.kernel_sizes[l], Cp, Cp); // conv [co][ci][k]
      tensor(C, 1, 1, Cp, 1); // conv bias
      tensor(C, 1, 1, Cp, 1); // input mixin [C][1]
      tensor(C, C, 1, Cp, Cp); // layer1x1 [co][ci]
      tensor(C, 1, 1, Cp, 1); // its bias
    }
    tensor(H, C, 1, Hp, Cp); // head rechannel [H][C]
    if (A.head_bias)
      tensor(H, 1, 1, Hp, 1);
  }
  out.weights.push_back(*(w++)); // head_scale
  return w == wn.weights.data() + wn.weights.size();
}

// Geometry of a (possibly downloaded, possibly corrupt) file before any 32-bit offset is derived from it: dilations and
// kernel sizes must be positive and the per-stream history — every ring is (K - 1) * dilation + 64 frames of its input
// channels, counted here with the channel padding the A1 kernels may add — must stay within 1 GiB, which also keeps
// every byte offset inside the kernels' 32-bit descriptors. The reference would throw std::bad_alloc or run out of
// memory on such a file; here it is a load error.
void validate_wavenet_geometry(const WaveNetSpec& wn)
{
  constexpr long long kMaxStateFloats = 1ll << 28; // 1 GiB of float32 per stream
  long long total = 0;
  auto ring = [&](long long K, long long dil, long long cin, const char* what) {
    if (K < 1 || dil < 1)
      throw std::runtime_error(std::string("plan: ") + what + " needs kernel_size >= 1 and dilation >= 1");
    if (K > 4096 || dil > (1ll << 26))
      throw std::runtime_error(std::string("plan: ") + what + " kernel_size / dilation out of range for the device path");
    const long long frames = (K - 1) * dil + kBlock;
    total += frames * ((cin + 3) / 4 * 4 + 4);
    if (total > kMaxStateFloats)
      throw std::runtime_error("plan: per-stream history exceeds 1 GiB (kernel_size x dilation too large for the device path)");
  };
  for (const LayerArraySpec& A : wn.arrays)
  {
    if (A.channels < 1 || A.channels > 4096 || A.bottleneck < 1 || A.bottleneck > 4096 || A.head_size < 1 || A.head_size > 4096)
      throw std::runtime_error("plan: channel counts out of range for the device path");
    for (int l = 0; l < A.num_layers(); l++)
      ring(A.kernel_sizes[(size_t)l], A.dilations[(size_t)l], A.channels, "a WaveNet layer");
    ring(A.head_kernel_size, A.head_dilation, A.head_output_size(), "a head rechannel");
  }
  if (wn.with_head)
    for (int k : wn.head.kernel_sizes)
      ring(k, 1, std::max(wn.head.channels, wn.head.in_channels), "a post-stack head convolution");
  if (wn.condition_dsp && wn.condition_dsp->arch == ARCH_WAVENET)
    validate_wavenet_geometry(wn.condition_dsp->wavenet);
}

} // namespace namhip
