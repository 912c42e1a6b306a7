// plan.cpp — ModelSpec -> device Plan (host side).
//
// The op program is a straight transcription of the reference's per-block control flow:
//   WaveNet::process            NAM/wavenet/model.cpp:822-910
//   LayerArray::Process(Inner)  NAM/wavenet/model.cpp:463-549
//   Layer::Process              NAM/wavenet/model.cpp:183-393
//   detail::Head::process       NAM/wavenet/model.cpp:86-103
// and the weight blob is built by walking the flat weight stream in set_weights_ order
// (model.cpp:152-181, 563-569, 661-683; Conv1D conv1d.cpp:40-55; Conv1x1 dsp.cpp:384-397).
#include "plan.h"
#include <cstdlib>
#include "kp_table.h"
#include "aq_table.h"

#include <algorithm>
#include <cstring>
#include <sstream>

namespace namhip
{
namespace
{

struct RowAlloc
{
  int top = 0;
  int high = 0;
  int alloc(int rows)
  {
    // every tensor owns a multiple of four rows: the kernel reads, computes and writes rows four at a time without
    // per-row tests (its weights / biases are zero-padded to match, so the padding rows hold zeros or finite scratch)
    rows = (rows + 3) / 4 * 4;
    const int r = top;
    top += rows;
    high = std::max(high, top);
    return r * kBlock; // float offset
  }
  int mark() const { return top; }
  void release(int m) { top = m; }
};

struct Builder
{
  Plan& plan;
  RowAlloc rows;
  int state_floats = 0; // ring area only; write-position table is prepended at the end

  explicit Builder(Plan& p)
  : plan(p)
  {
  }

  int blob_reserve(size_t n, size_t align = 16)
  {
    while (plan.blob.size() % align)
      plan.blob.push_back(0.0f);
    const int off = (int)plan.blob.size();
    plan.blob.resize(plan.blob.size() + n, 0.0f);
    return off;
  }

  NamOp& push(int type)
  {
    NamOp op;
    std::memset(&op, 0, sizeof(op));
    op.type = type;
    op.w = -1;
    op.b = -1;
    op.state = -1;
    plan.ops.push_back(op);
    return plan.ops.back();
  }

  // Dense conv (K >= 1) consuming weights from the flat stream. Returns nothing; dst rows must exist.
  void conv(const float*& w, int dst, int src, int cin, int cout, int K, int dil, int groups, bool bias)
  {
    // outputs are produced in register blocks of `cb` = pad4(cout) channels (one block up to 16 outputs)
    const int cb = std::min((cout + 3) / 4 * 4, 16);
    const int cout_pad = (cout + cb - 1) / cb * cb;
    const int cin_pad = (cin + 3) / 4 * 4; // zero rows: the kernel reads input channels four at a time
    const int woff = blob_reserve((size_t)K * cin_pad * cout_pad);
    const int opg = cout / groups, ipg = cin / groups;
    for (int g = 0; g < groups; g++)
      for (int i = 0; i < opg; i++)
        for (int j = 0; j < ipg; j++)
          for (int k = 0; k < K; k++)
            plan.blob[(size_t)woff + ((size_t)k * cin_pad + (g * ipg + j)) * cout_pad + (g * opg + i)] = *(w++);
    int boff = -1;
    if (bias)
    {
      boff = blob_reserve((size_t)cout_pad);
      for (int i = 0; i < cout; i++)
        plan.blob[(size_t)boff + i] = *(w++);
    }
    NamOp& op = push(OP_CONV);
    op.dst = dst;
    op.src = src;
    op.cin = cin;
    op.cout = cout;
    op.cout_pad = cout_pad;
    op.cb = cb;
    op.w = woff;
    op.b = boff;
    op.k = K;
    op.dil = dil;
    const int lookback = (K - 1) * dil;
    if (lookback > 0)
    {
      op.ring = lookback + kBlock;
      op.state = state_floats;
      op.ring_id = plan.n_rings++;
      state_floats += cin * op.ring;
      // small histories are copied to LDS once per block (OP_STAGE) instead of being read tap by tap from HBM
      if (cin * lookback <= kBlock && (int)stages.size() < kMaxStages)
      {
        plan.ops.back().flag = 4;
        plan.ops.back().aux = hist_floats; // relative to the history area; fixed up by finish_stages()
        StageRec r;
        r.conv_op = (int)plan.ops.size() - 1;
        r.hist = hist_floats;
        stages.push_back(r);
        hist_floats += (cin * lookback + 3) / 4 * 4;
      }
    }
  }

  // ---- history staging (OP_STAGE) ----
  static constexpr int kMaxStages = 16;
  struct StageRec
  {
    int conv_op, hist;
  };
  std::vector<StageRec> stages;
  int hist_floats = 0;
  // Called once the whole program is emitted: places the history area behind the activation rows, points the staged
  // convs at it and inserts the run of OP_STAGE ops behind OP_LOAD_IN (position `at`).
  void finish_stages(size_t at)
  {
    if (stages.empty())
      return;
    const int base = rows.high * kBlock;
    std::vector<NamOp> run;
    for (const StageRec& r : stages)
    {
      NamOp& c = plan.ops[(size_t)r.conv_op];
      c.aux = base + r.hist;
      NamOp s;
      std::memset(&s, 0, sizeof(s));
      s.type = OP_STAGE;
      s.w = s.b = -1;
      s.dst = base + r.hist;
      s.cin = c.cin;
      s.k = (c.k - 1) * c.dil; // lookback in frames
      s.state = c.state;
      s.ring = c.ring;
      s.ring_id = c.ring_id;
      run.push_back(s);
    }
    run[0].cout = (int)run.size();
    plan.ops.insert(plan.ops.begin() + (long)at, run.begin(), run.end());
    rows.high += (hist_floats + kBlock - 1) / kBlock;
  }

  // FiLM (film.h:76-204): scale/shift = Conv1x1(cond) + bias; dst = src * scale (+ shift)
  // Emitted as ONE op: a 1x1 OP_CONV over the condition rows whose epilogue applies the scale (and shift) to the
  // rows `src` instead of storing them (`flag` 1 = scale, 2 = scale + shift; `aux` = src). The output channels are
  // re-ordered so that a register block of `cb` accumulators holds the scales of cb/2 channels followed by their
  // shifts (all cb are scales without a shift): no scale/shift rows in LDS, no separate OP_FILM.
  void film(const float*& w, const FilmSpec& f, int dst, int src, int cond, int cond_dim, int dim)
  {
    const int cout = (f.shift ? 2 : 1) * dim;
    // channels per register block: pad4(dim), at most 8 with a shift (16 accumulators) / 16 without
    const int per = std::min((dim + 3) / 4 * 4, f.shift ? 8 : 16);
    const int cb = f.shift ? 2 * per : per;
    const int cout_pad = (dim + per - 1) / per * cb;
    auto col = [&](int o) {
      const int c = o < dim ? o : o - dim;
      return (c / per) * cb + (o < dim ? 0 : per) + c % per;
    };
    const int woff = blob_reserve((size_t)((cond_dim + 3) / 4 * 4) * cout_pad); // input rows padded to a multiple of 4
    const int opg = cout / f.groups, ipg = cond_dim / f.groups;
    for (int g = 0; g < f.groups; g++)
      for (int i = 0; i < opg; i++)
        for (int j = 0; j < ipg; j++)
          plan.blob[(size_t)woff + (size_t)(g * ipg + j) * cout_pad + col(g * opg + i)] = *(w++);
    const int boff = blob_reserve((size_t)cout_pad);
    for (int i = 0; i < cout; i++)
      plan.blob[(size_t)boff + col(i)] = *(w++);
    NamOp& op = push(OP_CONV);
    op.dst = dst;
    op.src = cond;
    op.aux = src;
    op.cin = cond_dim;
    op.cout = dim; // rows written
    op.cout_pad = cout_pad;
    op.cb = cb;
    op.w = woff;
    op.b = boff;
    op.k = 1;
    op.dil = 1;
    op.flag = f.shift ? 2 : 1;
  }

  int act_params(const ActSpec& a)
  {
    const int off = blob_reserve(4 + std::max<size_t>(a.slopes.size(), 1), 4);
    for (int i = 0; i < 4; i++)
      plan.blob[(size_t)off + i] = a.p[i];
    for (size_t i = 0; i < a.slopes.size(); i++)
      plan.blob[(size_t)off + 4 + i] = a.slopes[i];
    return off;
  }

  void act(const ActSpec& a, int buf, int channels)
  {
    if (a.type == ACT_IDENTITY)
      return;
    NamOp& op = push(OP_ACT);
    op.dst = buf;
    op.cout = channels;
    op.k = a.type;
    op.ring = (int)a.slopes.size();
    const int off = act_params(a);
    plan.ops.back().w = off;
  }

  void simple(int type, int dst, int src, int aux, int channels)
  {
    NamOp& op = push(type);
    op.dst = dst;
    op.src = src;
    op.aux = aux;
    op.cout = channels;
  }

  // Emits one WaveNet; returns the LDS offset of its (already head_scale-d) output rows.
  // `in_rows` holds the raw input (in_channels rows). Rows allocated for the result stay allocated.
  int wavenet(const WaveNetSpec& wn, int in_rows)
  {
    if ((long)wn.weights.size() != wn.expected_weight_count())
      throw std::runtime_error("plan: WaveNet weight count mismatch");
    const float* w = wn.weights.data();

    // _process_condition model.cpp:777-807
    int cond = in_rows;
    int cond_dim = wn.in_channels;
    if (wn.condition_dsp)
    {
      if (wn.condition_dsp->arch != ARCH_WAVENET)
        throw std::runtime_error("plan: condition_dsp must be a WaveNet for the device path");
      cond = wavenet(wn.condition_dsp->wavenet, in_rows);
      cond_dim = wn.condition_dsp->wavenet.out_channels();
    }

    int prev_layer_out = -1, prev_head_out = -1;
    for (size_t ai = 0; ai < wn.arrays.size(); ai++)
    {
      const LayerArraySpec& A = wn.arrays[ai];
      if (A.condition_size != cond_dim)
        throw std::runtime_error("plan: condition_size does not match the condition signal");
      const int C = A.channels, B = A.bottleneck, HO = A.head_output_size();
      // persistent rows for this array
      const int head_acc = rows.alloc(HO);
      const int xa = rows.alloc(C), xb = rows.alloc(C);
      const int head_out = rows.alloc(A.head_size);
      // head accumulator init — model.cpp:463-486
      if (ai == 0)
        simple(OP_ZERO, head_acc, 0, 0, HO);
      else
        simple(OP_COPY, head_acc, prev_head_out, 0, HO);
      // rechannel — model.cpp:492
      const int layer_in = (ai == 0) ? in_rows : prev_layer_out;
      conv(w, xa, layer_in, A.input_size, C, 1, 1, 1, false);
      int x = xa, xn = xb;
      for (int l = 0; l < A.num_layers(); l++)
      {
        const int m = rows.mark();
        const int gm = A.gating_modes[l];
        const int zc = gm != GATING_NONE ? 2 * B : B;
        // The flat stream order is conv, mixin, layer1x1, head1x1, then the 8 FiLMs (model.cpp:152-181),
        // which differs from execution order. Resolve the per-module stream positions first.
        const float* w_conv = w;
        const float* p = w_conv + ((long)A.kernel_sizes[l] * C * zc / A.groups_input + zc);
        const float* w_mix = p;
        p += (long)A.condition_size * zc / A.groups_input_mixin;
        const float* w_l1 = p;
        if (A.layer1x1_active)
          p += (long)B * C / A.layer1x1_groups + C;
        const float* w_h1 = p;
        if (A.head1x1_active)
          p += (long)B * A.head1x1_out / A.head1x1_groups + A.head1x1_out;
        const int dims[FILM_COUNT] = {C, zc, A.condition_size, zc, zc, B, C, A.head1x1_out};
        const float* w_film[FILM_COUNT];
        bool film_on[FILM_COUNT];
        for (int k = 0; k < FILM_COUNT; k++)
        {
          film_on[k] = A.film[k].active;
          if (k == FILM_LAYER1X1_POST && !A.layer1x1_active)
            film_on[k] = false;
          if (k == FILM_HEAD1X1_POST && !A.head1x1_active)
            film_on[k] = false;
          w_film[k] = p;
          if (film_on[k])
          {
            const int outc = (A.film[k].shift ? 2 : 1) * dims[k];
            p += (long)A.condition_size * outc / A.film[k].groups + outc;
          }
        }
        w = p; // next layer

        // conv (+ pre/post FiLM) — model.cpp:189-203
        const int conv_out = rows.alloc(zc);
        int conv_in = x;
        if (film_on[FILM_CONV_PRE])
        {
          conv_in = rows.alloc(C);
          film(w_film[FILM_CONV_PRE], A.film[FILM_CONV_PRE], conv_in, x, cond, cond_dim, C);
        }
        conv(w_conv, conv_out, conv_in, C, zc, A.kernel_sizes[l], A.dilations[l], A.groups_input, true);
        if (film_on[FILM_CONV_POST])
          film(w_film[FILM_CONV_POST], A.film[FILM_CONV_POST], conv_out, conv_out, cond, cond_dim, zc);
        // input mixin (+ pre/post FiLM) — model.cpp:205-219
        int mix_in = cond;
        if (film_on[FILM_MIXIN_PRE])
        {
          mix_in = rows.alloc(cond_dim);
          film(w_film[FILM_MIXIN_PRE], A.film[FILM_MIXIN_PRE], mix_in, cond, cond, cond_dim, cond_dim);
        }
        const int mix_out = rows.alloc(zc);
        conv(w_mix, mix_out, mix_in, cond_dim, zc, 1, 1, A.groups_input_mixin, false);
        if (film_on[FILM_MIXIN_POST])
          film(w_film[FILM_MIXIN_POST], A.film[FILM_MIXIN_POST], mix_out, mix_out, cond, cond_dim, zc);
        // z = conv + mixin — model.cpp:220 (z aliases conv_out)
        const int z = conv_out;
        simple(OP_ADD, z, conv_out, mix_out, zc);
        if (film_on[FILM_ACT_PRE])
          film(w_film[FILM_ACT_PRE], A.film[FILM_ACT_PRE], z, z, cond, cond_dim, zc);
        // activation + 1x1 — model.cpp:234-288
        int l1 = -1;
        if (gm == GATING_NONE)
          act(A.activations[l], z, zc);
        else
        {
          NamOp& op = push(OP_GATE);
          op.dst = z;
          op.cout = B;
          op.flag = gm;
          op.k = A.activations[l].type;
          op.dil = A.secondary_activations[l].type;
          op.ring = (int)A.activations[l].slopes.size();
          op.ring_id = (int)A.secondary_activations[l].slopes.size();
          const int o1 = act_params(A.activations[l]);
          const int o2 = act_params(A.secondary_activations[l]);
          plan.ops.back().w = o1;
          plan.ops.back().b = o2;
        }
        if (film_on[FILM_ACT_POST])
          film(w_film[FILM_ACT_POST], A.film[FILM_ACT_POST], z, z, cond, cond_dim, B);
        if (A.layer1x1_active)
        {
          l1 = rows.alloc(C);
          conv(w_l1, l1, z, B, C, 1, 1, A.layer1x1_groups, true);
          // quirk: layer1x1_post_film is applied in the BLENDED branch only — model.cpp:282-286
          if (gm == GATING_BLENDED && film_on[FILM_LAYER1X1_POST])
            film(w_film[FILM_LAYER1X1_POST], A.film[FILM_LAYER1X1_POST], l1, l1, cond, cond_dim, C);
        }
        // head contribution — model.cpp:290-352, accumulated at :513-531
        int head_src = z;
        if (A.head1x1_active)
        {
          head_src = rows.alloc(A.head1x1_out);
          conv(w_h1, head_src, z, B, A.head1x1_out, 1, 1, A.head1x1_groups, true);
          if (film_on[FILM_HEAD1X1_POST])
            film(w_film[FILM_HEAD1X1_POST], A.film[FILM_HEAD1X1_POST], head_src, head_src, cond, cond_dim,
                 A.head1x1_out);
        }
        simple(OP_ADD, head_acc, head_acc, head_src, HO);
        // residual — model.cpp:354-392
        if (A.layer1x1_active)
        {
          simple(OP_ADD, xn, x, l1, C);
          std::swap(x, xn);
        }
        rows.release(m);
      }
      // head rechannel (causal Conv1D) — model.cpp:547-548
      conv(w, head_out, head_acc, HO, A.head_size, A.head_kernel_size, A.head_dilation, 1, A.head_bias);
      prev_layer_out = x;
      prev_head_out = head_out;
    }

    const int hs = wn.arrays.back().head_size;
    int result;
    if (wn.with_head)
    {
      // model.cpp:854-883 + Head::process :86-103. head_scale itself is the last weight, after the head convs.
      const PostHeadSpec& H = wn.head;
      long head_w = 0;
      {
        int cin = H.in_channels;
        for (size_t i = 0; i < H.kernel_sizes.size(); i++)
        {
          const int cout = (i + 1 == H.kernel_sizes.size()) ? H.out_channels : H.channels;
          head_w += (long)H.kernel_sizes[i] * cin * cout + cout;
          cin = cout;
        }
      }
      const float head_scale = w[head_w];
      int work = rows.alloc(hs);
      const int so = blob_reserve(1, 1);
      plan.blob[(size_t)so] = head_scale;
      {
        NamOp& op = push(OP_SCALE);
        op.dst = work;
        op.src = prev_head_out;
        op.cout = hs;
        plan.ops.back().w = so;
      }
      int cin = H.in_channels;
      for (size_t i = 0; i < H.kernel_sizes.size(); i++)
      {
        const int cout = (i + 1 == H.kernel_sizes.size()) ? H.out_channels : H.channels;
        act(H.activation, work, cin);
        const int o = rows.alloc(cout);
        conv(w, o, work, cin, cout, H.kernel_sizes[i], 1, 1, true);
        work = o;
        cin = cout;
      }
      w++; // head_scale
      result = work;
    }
    else
    {
      const float head_scale = *(w++); // model.cpp:670 — last weight overrides the JSON head_scale
      const int so = blob_reserve(1, 1);
      plan.blob[(size_t)so] = head_scale;
      result = rows.alloc(hs);
      NamOp& op = push(OP_SCALE);
      op.dst = result;
      op.src = prev_head_out;
      op.cout = hs;
      plan.ops.back().w = so;
    }
    if (w != wn.weights.data() + wn.weights.size())
      throw std::runtime_error("plan: internal error, weight stream not fully consumed");
    return result;
  }
};

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

} // namespace

// --------------------------------------------------------------------------------------------
// nam_wn_reg_kernel: macro-op program + padded dense weights (plan.h: WrPlan)
// --------------------------------------------------------------------------------------------
namespace
{
struct WrBuilder
{
  WrPlan& wr;
  int hist = 0; // floats of ring area laid out so far (a multiple of 4)
  struct Entry // one 64-frame window of one channel (plan.h: tables)
  {
    int32_t off, ring, slot_gs, o;
  };
  std::vector<Entry> rows, pf;
  std::vector<int32_t> ring_of_slot;
  // where shapes are looked up: the ahead-of-time tables (nullptr), or the model's own shape set (per-model compile)
  WrShapeSet* dyn = nullptr;
  enum Policy
  {
    AOT_EXACT_ONLY, // only fully described ahead-of-time shapes (and runs / pairs)
    AOT_ANY, // run-time-flag instantiations too
    JIT // register every shape in `dyn`
  } policy = AOT_ANY;
  WrBuilder(WrPlan& w, Policy p, WrShapeSet* d) : wr(w), dyn(d), policy(p) {}
  int shape_layer(int cond, int C, int B, bool G, int K, int HO, int flags, int act, int act2, bool l1)
  {
    if (policy == JIT)
      return dyn->layer(cond, C, B, G, K, HO, flags, act, act2, l1);
    const int id = wr_layer_shape(cond, C, B, G, K, HO, flags, act, act2, l1);
    return (id >= 0 && policy == AOT_EXACT_ONLY && !wr_layer_shape_is_exact(id)) ? -1 : id;
  }
  int shape_run(int C, int act) { return policy == JIT ? dyn->run(C, act) : wr_run_shape(C, act); }
  int shape_pair(int n_in, int n_out) { return policy == JIT ? dyn->pair(n_in, n_out) : wr_pair_shape(n_in, n_out); }

  // A layer's conv-input ring: [ceil(C / 4)][R][gs] floats; table entries for its channels. Returns the float offset
  // of the area (relative to the ring area's start).
  int ring_area(int C, int K, int dil)
  {
    const long lookback = (long)(K - 1) * dil;
    if (lookback + kBlock > (1 << 20))
      throw Unsupported("a conv reaching more than 2^20 frames back");
    const int R = (int)lookback + kBlock;
    const int slot = (int)ring_of_slot.size();
    if (slot >= kWrPosInts)
      throw Unsupported("more than 64 layers");
    ring_of_slot.push_back(R);
    const int off = hist;
    // the offsets (1 = the frame before the block) a block's taps can reach: tap L reads frames t - L, t = 0 .. 63
    std::vector<char> need((size_t)lookback + 1, 0);
    for (int k = 0; k + 1 < K; k++)
    {
      const long L = (long)(K - 1 - k) * dil;
      for (long o = std::max(1l, L - (kBlock - 1)); o <= L; o++)
        need[(size_t)o] = 1;
    }
    std::vector<int> windows; // o of lane 0; a window covers offsets o - 63 .. o (offsets < 1 land in the block being written)
    for (long hi = lookback; hi >= 1;)
    {
      if (!need[(size_t)hi])
      {
        hi--;
        continue;
      }
      const long o = std::max<long>(hi, kBlock);
      windows.push_back((int)o);
      hi = o - kBlock;
    }
    for (int q = 0; q * 4 < C; q++)
    {
      const int gs = std::min(4, C - 4 * q);
      for (int i = 0; i < gs; i++)
      {
        const int32_t eo = off + q * 4 * R + i;
        rows.push_back({eo, R, slot | (gs << 8), 0});
        for (int o : windows)
          pf.push_back({eo, R, slot | (gs << 8), o});
      }
    }
    hist += wr_pad4(C * R);
    return off;
  }
  int table(const std::vector<Entry>& t)
  {
    const int off = reserve((int)t.size() * 4);
    if (!t.empty())
      std::memcpy(&wr.blob[(size_t)off], t.data(), t.size() * sizeof(Entry));
    return off;
  }

  struct Unsupported : std::runtime_error
  {
    using std::runtime_error::runtime_error;
  };

  int reserve(int n)
  {
    const int off = (int)wr.blob.size();
    wr.blob.resize((size_t)off + (size_t)wr_pad4(n), 0.0f);
    return off;
  }
  WrOp& push(int type)
  {
    WrOp op;
    std::memset(&op, 0, sizeof(op));
    op.type = type;
    op.shape = -1;
    wr.ops.push_back(op);
    return wr.ops.back();
  }
  // dense, transposed [K * cin][pad4(cout)] at `dst` (row = tap * cin + in), from the reference's stream order
  // (groups, out, in, tap); `out0` / `out_n`: only outputs [out0, out0 + out_n) of the stream's `cout` land here, as
  // columns 0.. (a FiLM's scale and shift halves are two matrices)
  void dense(float* dst, const float*& w, int cin, int cout, int K, int groups, int out0 = 0, int out_n = -1, bool advance = true)
  {
    if (out_n < 0)
      out_n = cout;
    const int row = wr_pad4(out_n);
    const int opg = cout / groups, ipg = cin / groups;
    const float* p = w;
    for (int g = 0; g < groups; g++)
      for (int i = 0; i < opg; i++)
        for (int j = 0; j < ipg; j++)
          for (int k = 0; k < K; k++, p++)
          {
            const int o = g * opg + i - out0;
            if (o >= 0 && o < out_n)
              dst[(size_t)(k * cin + g * ipg + j) * row + o] = *p;
          }
    if (advance)
      w = p;
  }
  void act_block(float* dst, const ActSpec& a, int rows_n)
  {
    for (int i = 0; i < 4; i++)
      dst[i] = a.p[i];
    if (a.type == ACT_PRELU && !a.slopes.empty())
      for (int c = 0; c < 16; c++)
        dst[4 + c] = a.slopes[(size_t)c % a.slopes.size()];
    (void)rows_n;
  }

  void net(const WaveNetSpec& wn, bool nested)
  {
    if (wn.with_head && (policy != JIT || !dyn || nested))
      throw Unsupported("a post-stack head (no ahead-of-time shapes: it needs the per-model compile)");
    if (wn.in_channels > kWrRegs || wn.out_channels() > kWrRegs)
      throw Unsupported("more than 8 input / output channels");
    if ((long)wn.weights.size() != wn.expected_weight_count())
      throw std::runtime_error("plan: WaveNet weight count mismatch");
    int cond_dim = wn.in_channels;
    if (wn.condition_dsp)
    {
      if (nested)
        throw Unsupported("a condition_dsp inside a condition_dsp");
      if (wn.condition_dsp->arch != ARCH_WAVENET)
        throw Unsupported("a condition_dsp that is not a WaveNet");
      const WaveNetSpec& c = wn.condition_dsp->wavenet;
      if (c.in_channels != wn.in_channels)
        throw Unsupported("a condition_dsp with another input width");
      net(c, true);
      cond_dim = c.out_channels();
      WrOp& op = push(WR_SET_COND);
      op.n_out = cond_dim;
      op.scale = c.weights.back(); // model.cpp:670 — the last weight is the head scale
    }
    const float* w = wn.weights.data();
    for (size_t ai = 0; ai < wn.arrays.size(); ai++)
    {
      const LayerArraySpec& A = wn.arrays[ai];
      const int C = A.channels, B = A.bottleneck, HO = A.head_output_size();
      if (A.condition_size != cond_dim)
        throw std::runtime_error("plan: condition_size does not match the condition signal");
      if (!A.layer1x1_active && policy != JIT)
        throw Unsupported("a layer without its 1x1"); // (compiled per model: the ahead-of-time shapes all have one)
      if (A.head_kernel_size != 1 && policy != JIT)
        throw Unsupported("a head rechannel with a kernel"); // (compiled per model only)
      if (ai > 0 && wn.arrays[ai - 1].head_size != HO)
        throw std::runtime_error("plan: head sizes of consecutive arrays do not chain");
      {
        WrOp& op = push(WR_ARRAY_BEGIN);
        op.flags = ai == 0 ? 1 : 0;
        op.n_in = A.input_size;
        op.n_out = C;
        op.shape = shape_pair(A.input_size, C);
        if (op.shape < 0)
          throw Unsupported("a rechannel of " + std::to_string(A.input_size) + " -> " + std::to_string(C));
        const int off = reserve(A.input_size * wr_pad4(C));
        wr.ops.back().w = off;
        dense(&wr.blob[(size_t)off], w, A.input_size, C, 1, 1);
      }
      for (int l = 0; l < A.num_layers(); l++)
      {
        const int gm = A.gating_modes[l];
        const bool G = gm != GATING_NONE;
        const int zc = G ? 2 * B : B, K = A.kernel_sizes[l], dil = A.dilations[l];
        const int h1o = A.head1x1_active ? A.head1x1_out : 0;
        const ActSpec& a1 = A.activations[l];
        const ActSpec& a2 = A.secondary_activations[l];
        if (a1.type == ACT_LUT || (G && a2.type == ACT_LUT))
          throw Unsupported("a look-up-table activation");
        // Activation::apply on the flat buffer indexes PReLU slopes by frame * rows + row (activations.h:283-297):
        // only frame-independent when the slope count divides the row count
        if (!G && a1.type == ACT_PRELU && !a1.slopes.empty() && zc % (int)a1.slopes.size() != 0)
          throw Unsupported("a PReLU whose slope count does not divide the channel count");
        if (zc > 16 || C > kWrRegs || HO > kWrRegs || cond_dim > kWrRegs)
          throw Unsupported("a layer wider than the register files");
        // the layer keeps its whole conv matrix in registers while the taps arrive ([K * C][pad4(zc)] floats per lane)
        // (its taps stay in registers for the whole layer: K * C floats per lane)
        if (K * C > 64)
          throw Unsupported("a conv of more than 64 tap inputs (kernel size " + std::to_string(K) + " x " + std::to_string(C) + " channels)");
        int flags = gm == GATING_BLENDED ? (1 << 16) : 0;
        for (int k = 0; k < FILM_COUNT; k++)
          if (A.film[k].active && !(k == FILM_HEAD1X1_POST && !A.head1x1_active))
            flags |= (1 << k) | (A.film[k].shift ? 1 << (8 + k) : 0);
        // a PLAIN layer (no gating, FiLM or head1x1; condition size 1, kernel size 3, at most four channels, a
        // parameterless activation) joins a WR_RUN and takes the compact weight block
        const bool plain = cond_dim == 1 && B == C && C <= 4 && !G && K == 3 && h1o == 0 && flags == 0 && A.layer1x1_active
                           && (a1.type == ACT_RELU || a1.type == ACT_TANH || a1.type == ACT_FASTTANH);
        const int run_shape = plain ? shape_run(C, a1.type) : -1;
        const int shape = run_shape >= 0 ? -1
                                         : shape_layer(cond_dim, C, B, G, K, h1o, flags, a1.type, G ? a2.type : (int)ACT_IDENTITY,
                                                       A.layer1x1_active);
        if (shape < 0 && run_shape < 0)
          throw Unsupported("layer shape cond=" + std::to_string(cond_dim) + " C=" + std::to_string(C) + " B=" + std::to_string(B)
                            + (G ? " gating" : "") + " K=" + std::to_string(K) + " head1x1=" + std::to_string(h1o));
        int off = 0;
        int film_matrix_floats = 0; // this layer's FiLM weights that lie in the matrix form (half the instructions per weight)
        if (run_shape >= 0)
        {
          const WrPlainLayout P = wr_plain_layout(C);
          off = reserve(P.total);
          float* d = &wr.blob[(size_t)off];
          // conv and layer1x1 in the matrix form: row `o` of the block = output o's weights over the inputs in order (zero rows
          // for o >= C, zero columns behind the last input: never multiplied)
          float t[12 * 4] = {0};
          dense(t, w, C, C, 3, A.groups_input); // [tap * C + channel][4 outputs]
          const int in4 = wr_pad4(3 * C);
          for (int o = 0; o < 4; o++)
            for (int j = 0; j < 3 * C; j++)
              d[P.conv + o * in4 + j] = t[j * 4 + o];
          for (int i = 0; i < C; i++)
            d[P.conv_b + i] = *(w++);
          dense(d + P.mixin, w, 1, C, 1, A.groups_input_mixin);
          std::fill(t, t + 16, 0.0f);
          dense(t, w, C, C, 1, A.layer1x1_groups);
          for (int o = 0; o < 4; o++)
            for (int j = 0; j < C; j++)
              d[P.l1 + o * 4 + j] = t[j * 4 + o];
          for (int i = 0; i < C; i++)
            d[P.l1_b + i] = *(w++);
        }
        else
        {
          const WrLayerLayout L = wr_layer_layout(cond_dim, C, B, G, K, h1o);
          off = reserve(L.total);
          float* d = &wr.blob[(size_t)off];
          // the flat stream order is conv, mixin, layer1x1, head1x1, then the 8 FiLMs (model.cpp:152-181)
          // [in][pad4(out)] (dense) -> the matrix form [output row % 4][quad][pad4(in)] (kernel_wn_reg.hip: WrMatM)
          std::vector<float> tm;
          auto matrix_form = [&](float* dst, int in_n, int out_n, int k_taps, int groups) {
            const int o4 = wr_pad4(out_n), i4 = wr_pad4(in_n), Q = o4 / 4;
            tm.assign((size_t)k_taps * in_n * o4, 0.0f);
            dense(tm.data(), w, in_n, out_n, k_taps, groups); // [tap * in_n + input][o4]
            for (int k = 0; k < k_taps; k++)
              for (int cls = 0; cls < 4; cls++)
                for (int q = 0; q < Q; q++)
                  for (int c = 0; c < in_n; c++)
                    dst[(size_t)k * o4 * i4 + (size_t)(cls * Q + q) * i4 + c] = tm[(size_t)(k * in_n + c) * o4 + 4 * q + cls];
          };
          matrix_form(d + L.conv, C, zc, K, A.groups_input);
          for (int i = 0; i < zc; i++)
            d[L.conv_b + i] = *(w++);
          dense(d + L.mixin, w, cond_dim, zc, 1, A.groups_input_mixin);
          if (A.layer1x1_active)
          {
            matrix_form(d + L.l1, B, C, 1, A.layer1x1_groups);
            for (int i = 0; i < C; i++)
              d[L.l1_b + i] = *(w++);
          }
          if (A.head1x1_active)
          {
            matrix_form(d + L.h1, B, h1o, 1, A.head1x1_groups);
            for (int i = 0; i < h1o; i++)
              d[L.h1_b + i] = *(w++);
          }
          const int dims[FILM_COUNT] = {C, zc, cond_dim, zc, zc, B, C, h1o};
          for (int k = 0; k < FILM_COUNT; k++)
          {
            bool on = A.film[k].active;
            if (k == FILM_HEAD1X1_POST && !A.head1x1_active)
              on = false;
            if (!on)
              continue;
            const int D = dims[k], outc = (A.film[k].shift ? 2 : 1) * D, D4 = wr_pad4(D);
            // Conv1x1(cond -> outc, groups) + bias; outputs [0, D) scale, [D, 2D) shift: two matrices, two bias vectors
            dense(d + L.film[k], w, cond_dim, outc, 1, A.film[k].groups, 0, D, !A.film[k].shift);
            if (A.film[k].shift)
              dense(d + L.film[k] + cond_dim * D4, w, cond_dim, outc, 1, A.film[k].groups, D, D);
            if (wr_film_matrix_form(cond_dim))
            {
              film_matrix_floats += (A.film[k].shift ? 2 : 1) * cond_dim * D4;
              // [cond][pad4(D)] -> [lane class][output quad][cond]: class i of quad q = row 4 q + i, its weights for inputs 0 .. cond - 1
              const int Q = D4 / 4;
              std::vector<float> t((size_t)cond_dim * D4);
              for (int m = 0; m < (A.film[k].shift ? 2 : 1); m++)
              {
                float* mat = d + L.film[k] + m * cond_dim * D4;
                std::copy(mat, mat + cond_dim * D4, t.begin());
                for (int cls = 0; cls < 4; cls++)
                  for (int q = 0; q < Q; q++)
                    for (int c = 0; c < cond_dim; c++)
                      mat[(cls * Q + q) * cond_dim + c] = t[(size_t)c * D4 + 4 * q + cls];
              }
            }
            float* bias = d + L.film[k] + 2 * cond_dim * D4;
            for (int i = 0; i < D; i++)
              bias[i] = *(w++);
            if (A.film[k].shift)
              for (int i = 0; i < D; i++)
                bias[D4 + i] = *(w++);
          }
          act_block(d + L.act, a1, zc);
          if (G)
            act_block(d + L.act2, a2, B);
        }
        WrOp& op = push(WR_LAYER);
        op.shape = shape;
        op.w = off;
        op.pad[0] = zc + (G ? B : 0); // (planner only, like pad[1]: wr_program_cuts — activation evaluations per frame)
        op.pad[1] = film_matrix_floats;
        op.slot = (int)ring_of_slot.size();
        op.run = run_shape + 1;
        op.hist = ring_area(C, K, dil); // + the ring area's base, added once the weights and tables are complete
        op.ring = (K - 1) * dil + kBlock;
        op.dil = dil;
        op.flags = flags;
        op.act = a1.type;
        op.act2 = G ? a2.type : ACT_IDENTITY;
        wr.n_layers++;
      }
      if (A.head_kernel_size == 1)
      {
        WrOp& op = push(WR_ARRAY_END);
        op.flags = A.head_bias ? 1 : 0;
        op.n_in = HO;
        op.n_out = A.head_size;
        op.shape = shape_pair(HO, A.head_size);
        if (op.shape < 0)
          throw Unsupported("a head rechannel of " + std::to_string(HO) + " -> " + std::to_string(A.head_size));
        const int off = reserve(HO * wr_pad4(A.head_size) + wr_pad4(A.head_size));
        wr.ops.back().w = off;
        dense(&wr.blob[(size_t)off], w, HO, A.head_size, 1, 1);
        if (A.head_bias)
          for (int i = 0; i < A.head_size; i++)
            wr.blob[(size_t)off + (size_t)HO * wr_pad4(A.head_size) + i] = *(w++);
      }
      else
      {
        // a Conv1D over the head accumulator: [K_h * HO][pad4(head size)] (row = tap * HO + input) + bias, its own ring
        const int KH = A.head_kernel_size;
        if (KH * HO > 64 || KH * HO * wr_pad4(A.head_size) > 320)
          throw Unsupported("a head rechannel of more than 64 tap inputs / 320 weights");
        const int off = reserve(KH * HO * wr_pad4(A.head_size) + wr_pad4(A.head_size));
        dense(&wr.blob[(size_t)off], w, HO, A.head_size, KH, 1);
        if (A.head_bias)
          for (int i = 0; i < A.head_size; i++)
            wr.blob[(size_t)off + (size_t)KH * HO * wr_pad4(A.head_size) + i] = *(w++);
        WrOp& op = push(WR_ARRAY_END_K);
        op.flags = A.head_bias ? 1 : 0;
        op.n_in = HO;
        op.n_out = A.head_size;
        op.shape = dyn->head(HO, A.head_size, KH);
        op.w = off;
        op.slot = (int)ring_of_slot.size();
        op.hist = ring_area(HO, KH, A.head_dilation); // + the ring area's base, below
        op.ring = (KH - 1) * A.head_dilation + kBlock;
        op.dil = A.head_dilation;
        wr.n_layers++; // (a slot: one write position per ring)
      }
    }
    // the post-stack head (model.cpp:21-103, applied :854-866): activation + Conv1D per entry of kernel_sizes, on the
    // last array's head output times head_scale; head_scale itself follows the head's weights in the stream
    size_t first_post = 0;
    if (wn.with_head)
    {
      const PostHeadSpec& H = wn.head;
      if (H.in_channels != wn.arrays.back().head_size || H.kernel_sizes.empty())
        throw Unsupported("a post-stack head whose input is not the last array's head output");
      first_post = wr.ops.size();
      int cin = H.in_channels;
      for (size_t i = 0; i < H.kernel_sizes.size(); i++)
      {
        const int cout = (i + 1 == H.kernel_sizes.size()) ? H.out_channels : H.channels;
        const int K = H.kernel_sizes[i];
        if (cin > kWrRegs || cout > kWrRegs || K * cin > 64 || K * cin * wr_pad4(cout) > 320)
          throw Unsupported("a post-stack head layer of more than 8 channels / 64 tap inputs / 320 weights");
        if (H.activation.type == ACT_LUT)
          throw Unsupported("a lookup-table activation in the post-stack head");
        const int off = reserve(K * cin * wr_pad4(cout) + wr_pad4(cout) + kWrActFloats);
        dense(&wr.blob[(size_t)off], w, cin, cout, K, 1);
        for (int o = 0; o < cout; o++) // Conv1D bias (always: set_size_(cin, cout, k, true, 1, 1))
          wr.blob[(size_t)off + (size_t)K * cin * wr_pad4(cout) + o] = *(w++);
        act_block(&wr.blob[(size_t)off + (size_t)K * cin * wr_pad4(cout) + wr_pad4(cout)], H.activation, cin);
        WrOp& op = push(WR_POST_HEAD);
        op.n_in = cin;
        op.n_out = cout;
        op.shape = dyn->post(cin, cout, K, H.activation.type);
        op.w = off;
        op.act = H.activation.type;
        op.scale = 1.0f;
        op.dil = 1;
        if (K > 1)
        {
          op.slot = (int)ring_of_slot.size();
          op.hist = ring_area(cin, K, 1); // + the ring area's base, below
          op.ring = (K - 1) + kBlock;
          wr.n_layers++;
        }
        cin = cout;
      }
    }
    const float head_scale = *(w++);
    if (w != wn.weights.data() + wn.weights.size())
      throw std::runtime_error("plan: internal error, weight stream not fully consumed (register-resident plan)");
    if (wn.with_head)
      wr.ops[first_post].scale = head_scale;
    if (!nested)
    {
      WrOp& op = push(WR_OUTPUT);
      op.n_out = wn.out_channels();
      op.scale = wn.with_head ? 1.0f : head_scale;
    }
  }
};
} // namespace

int wr_layer_shape(int cond, int channels, int bottleneck, bool gating, int kernel, int head_out, int flags, int act,
                   int act2, bool l1)
{
  // `flags` as in WrOp::flags: bits 0-7 FiLM slots, 8-15 their shifts, bit 16 blended
#define X(ID, COND, C, B, G, K, HO, FM, SM, BL, A1, A2, L1) \
  if (cond == COND && channels == C && bottleneck == B && gating == G && kernel == K && head_out == HO && l1 == (L1 != 0) \
      && (FM < 0 || (flags == (FM | (SM << 8) | (BL << 16)) && act == A1 && act2 == A2))) \
    return ID;
  WR_LAYER_SHAPES(X)
#undef X
  return -1;
}

bool wr_layer_shape_is_exact(int id)
{
#define X(ID, COND, C, B, G, K, HO, FM, SM, BL, A1, A2, L1) \
  if (id == ID) \
    return FM >= 0;
  WR_LAYER_SHAPES(X)
#undef X
  return false;
}

int wr_run_shape(int channels, int act)
{
#define X(ID, C, A) \
  if (channels == C && act == A) \
    return ID;
  WR_RUN_SHAPES(X)
#undef X
  return -1;
}

int wr_pair_shape(int n_in, int n_out)
{
#define X(ID, IN, OUT) \
  if (n_in == IN && n_out == OUT) \
    return ID;
  WR_PAIR_SHAPES(X)
#undef X
  return -1;
}

int WrShapeSet::layer(int cond, int C, int B, bool G, int K, int HO, int flags, int act, int act2, bool l1)
{
  const Layer want{cond, C, B, G ? 1 : 0, K, HO, flags, act, act2, l1 ? 1 : 0};
  for (size_t i = 0; i < layers.size(); i++)
  {
    const Layer& o = layers[i];
    if (o.cond == want.cond && o.C == want.C && o.B == want.B && o.G == want.G && o.K == want.K && o.HO == want.HO
        && o.flags == want.flags && o.act == want.act && o.act2 == want.act2 && o.l1 == want.l1)
      return (int)i;
  }
  layers.push_back(want);
  return (int)layers.size() - 1;
}
int WrShapeSet::run(int C, int act)
{
  for (size_t i = 0; i < runs.size(); i++)
    if (runs[i].C == C && runs[i].act == act)
      return (int)i;
  runs.push_back({C, act});
  return (int)runs.size() - 1;
}
int WrShapeSet::post(int n_in, int n_out, int K, int act)
{
  for (size_t i = 0; i < posts.size(); i++)
    if (posts[i].n_in == n_in && posts[i].n_out == n_out && posts[i].K == K && posts[i].act == act)
      return (int)i;
  posts.push_back({n_in, n_out, K, act});
  return (int)posts.size() - 1;
}
int WrShapeSet::head(int n_in, int n_out, int K)
{
  for (size_t i = 0; i < heads.size(); i++)
    if (heads[i].n_in == n_in && heads[i].n_out == n_out && heads[i].K == K)
      return (int)i;
  heads.push_back({n_in, n_out, K});
  return (int)heads.size() - 1;
}
int WrShapeSet::pair(int n_in, int n_out)
{
  for (size_t i = 0; i < pairs.size(); i++)
    if (pairs[i].n_in == n_in && pairs[i].n_out == n_out)
      return (int)i;
  pairs.push_back({n_in, n_out});
  return (int)pairs.size() - 1;
}
std::string WrShapeSet::header_text() const
{
  // the tables of plan.h, generated: every layer fully described (FiLM set, blend, activation types compiled in)
  std::stringstream ss;
  ss << "#define NAM_WR_JIT_SHAPES 1\n#define WR_LAYER_SHAPES(X)";
  for (size_t i = 0; i < layers.size(); i++)
  {
    const Layer& o = layers[i];
    ss << " X(" << i << ", " << o.cond << ", " << o.C << ", " << o.B << ", " << (o.G ? "true" : "false") << ", " << o.K << ", " << o.HO
       << ", " << (o.flags & 0xff) << ", " << ((o.flags >> 8) & 0xff) << ", " << ((o.flags >> 16) & 1) << ", " << o.act << ", " << o.act2
       << ", " << o.l1 << ")";
  }
  ss << "\n#define WR_RUN_SHAPES(X)";
  for (size_t i = 0; i < runs.size(); i++)
    ss << " X(" << i << ", " << runs[i].C << ", " << runs[i].act << ")";
  ss << "\n#define WR_PAIR_SHAPES(X)";
  for (size_t i = 0; i < pairs.size(); i++)
    ss << " X(" << i << ", " << pairs[i].n_in << ", " << pairs[i].n_out << ")";
  ss << "\n#define WR_HEADK_SHAPES(X)";
  for (size_t i = 0; i < heads.size(); i++)
    ss << " X(" << i << ", " << heads[i].n_in << ", " << heads[i].n_out << ", " << heads[i].K << ")";
  ss << "\n#define WR_POSTHEAD_SHAPES(X)";
  for (size_t i = 0; i < posts.size(); i++)
    ss << " X(" << i << ", " << posts[i].n_in << ", " << posts[i].n_out << ", " << posts[i].K << ", " << posts[i].act << ")";
  ss << "\n";
  size_t total_ops = 0, max_ops = 1;
  for (const auto& pr : programs)
  {
    total_ops += pr.ops.size() + pr.ops_cut.size();
    max_ops = std::max(max_ops, std::max(pr.ops.size(), pr.ops_cut.size()));
  }
  if (!programs.empty() && total_ops <= 1536) // (a model of hundreds of ops stays a walked program: code size)
  {
    ss << "#define NAM_WR_PROGRAMS 1\n#define NAM_WR_N_PROGRAMS " << programs.size() << "\n#define NAM_WR_MAX_OPS " << max_ops << "\n";
    ss << "#define NAM_WR_PROGRAM_SPLITS {";
    for (size_t i = 0; i < programs.size(); i++)
      ss << (i ? ", " : "") << "{" << programs[i].split_op[0] << ", " << programs[i].split_op[1] << ", " << programs[i].split_op[2] << ", "
         << programs[i].split_op[3] << "}";
    ss << "}\n";
    for (int cut = 0; cut < 2; cut++)
    {
      ss << "#define " << (cut ? "NAM_WR_PROGRAM_COUNTS_CUT" : "NAM_WR_PROGRAM_COUNTS") << " {";
      for (size_t i = 0; i < programs.size(); i++)
        ss << (i ? ", " : "") << (cut ? programs[i].ops_cut : programs[i].ops).size();
      ss << "}\n#define " << (cut ? "NAM_WR_PROGRAM_OPS_CUT" : "NAM_WR_PROGRAM_OPS") << " {";
      for (size_t i = 0; i < programs.size(); i++)
      {
        const std::vector<WrOp>& ops = cut ? programs[i].ops_cut : programs[i].ops;
        ss << (i ? ", " : "") << "{";
        for (size_t k = 0; k < max_ops; k++)
        {
          WrOp o;
          std::memset(&o, 0, sizeof(o));
          if (k < ops.size())
            o = ops[k];
          int32_t scale_bits;
          std::memcpy(&scale_bits, &o.scale, sizeof(scale_bits));
          // {type, shape, w, hist, ring, dil, flags, act, act2, n_in, n_out, scale_bits, slot}; a WR_RUN's slot = its first record
          const int32_t slot = o.type == WR_RUN ? programs[i].first_rec + o.pad[0] : o.slot;
          ss << (k ? ", " : "") << "{" << o.type << ", " << o.shape << ", " << o.w << ", " << o.hist << ", " << o.ring << ", " << o.dil << ", "
             << o.flags << ", " << o.act << ", " << o.act2 << ", " << o.n_in << ", " << o.n_out << ", " << scale_bits << ", " << slot << "}";
        }
        ss << "}";
      }
      ss << "}\n";
    }
    ss << "#define NAM_WR_RUN_RECS {";
    for (size_t i = 0; i < run_recs.size(); i++)
      ss << (i ? ", " : "") << "{" << run_recs[i][0] << ", " << run_recs[i][1] << ", " << run_recs[i][2] << ", " << run_recs[i][3] << "}";
    if (run_recs.empty())
      ss << "{0, 0, 0, 0}";
    ss << "}\n";
  }
  return ss.str();
}

// Two- / four-stage launches (kernel_wn_reg.hip, NST) cut the program where the work balances; an op's weights are a fair
// measure of its arithmetic (every weight is one multiply-add per frame): op i owns the blob from its offset to the next
// larger one (`weights_end`: the first table behind the weights). split[q], q = 0 .. 2 = the cut closest to (q + 1) / 4 of the work
// (four wavefronts per stream); split[3] = the TWO-wave cut, which also counts an activation evaluation as sixteen weights (ten
// vector instructions, two of them at a quarter of the rate: a gated 12-row layer of a condition_dsp is a third activations) —
// calibrated on config 4, same-box: the second of two waves from op 5 / 6 / 7 / 8 on reads 4.49 / 4.23 / 4.58 / 5.84 us per step
// (the term picks 6); with the same term the four-wave cuts become {2, 6, 11} and 256 streams read 3.68 us instead of 3.14 for
// {2, 7, 12}: four short parts are dominated by their matrix work, two long ones are not.
static void wr_program_cuts(const std::vector<WrOp>& ops, int weights_end, int split[4])
{
  auto weighs = [](const WrOp& op) {
    return op.type == WR_LAYER || op.type == WR_RUN || op.type == WR_ARRAY_BEGIN || op.type == WR_ARRAY_END || op.type == WR_ARRAY_END_K
           || op.type == WR_POST_HEAD;
  };
  std::vector<int> ws;
  for (const auto& op : ops)
    if (weighs(op))
      ws.push_back(op.w);
  ws.push_back(weights_end);
  std::sort(ws.begin(), ws.end());
  std::vector<long> cost(ops.size(), 8);
  long total = 0;
  for (size_t i = 0; i < ops.size(); i++)
  {
    const auto& op = ops[i];
    if (weighs(op))
    {
      const auto nx = std::upper_bound(ws.begin(), ws.end(), op.w);
      cost[i] += nx != ws.end() ? *nx - op.w : 0;
      // a FiLM matrix in the matrix form (kernel_wn_reg.hip: WrFilm) costs one matrix instruction per four weights and one LDS
      // read per sixteen, against one packed FMA per two and one read per four: 0.45 of its weights
      if (op.type == WR_LAYER)
        cost[i] -= (long)op.pad[1] * 55 / 100;
    }
    total += cost[i];
  }
  auto cut_at = [&](const std::vector<long>& c, long tot, int num, int den) { // the cut closest to num / den of the work
    long acc = 0, best = -1;
    int at = 0;
    for (size_t m = 1; m < ops.size(); m++)
    {
      acc += c[m - 1];
      const long d = std::labs(den * acc - num * tot);
      if (best < 0 || d < best)
      {
        best = d;
        at = (int)m;
      }
    }
    return at;
  };
  for (int q = 0; q < 3; q++)
    split[q] = cut_at(cost, total, q + 1, 4);
  std::vector<long> cost2 = cost;
  long total2 = total;
  for (size_t i = 0; i < ops.size(); i++)
    if (ops[i].type == WR_LAYER)
    {
      cost2[i] += 16l * ops[i].pad[0];
      total2 += 16l * ops[i].pad[0];
    }
  split[3] = cut_at(cost2, total2, 1, 2);
}

// One attempt under one shape policy; throws WrBuilder::Unsupported
static void build_wr_with(const WaveNetSpec& wn, WrPlan& wr, WrBuilder::Policy policy, WrShapeSet* dyn)
{
  {
    WrBuilder b(wr, policy, dyn);
    b.net(wn, false);
    static_assert(sizeof(WrBuilder::Entry) == 16, "table entries are int4");
    // consecutive plain layers of one shape (weight blocks at the layout's stride) become one WR_RUN
    struct Run
    {
      size_t op; // index of the WR_RUN op
      int table; // blob float offset of its records
      std::vector<WrOp> layers;
    };
    std::vector<Run> runs;
    {
      std::vector<WrOp> fused;
      for (size_t i = 0; i < wr.ops.size();)
      {
        const WrOp& o = wr.ops[i];
        if (o.type != WR_LAYER || o.run <= 0)
        {
          fused.push_back(o);
          i++;
          continue;
        }
        size_t j = i + 1;
        while (j < wr.ops.size() && wr.ops[j].type == WR_LAYER && wr.ops[j].run == o.run
               && wr.ops[j].w - wr.ops[j - 1].w == wr.ops[i + 1].w - o.w)
          j++;
        WrOp r;
        std::memset(&r, 0, sizeof(r));
        r.type = WR_RUN;
        r.shape = o.run - 1;
        r.w = o.w;
        r.n_in = (int)(j - i);
        r.n_out = j - i > 1 ? wr.ops[i + 1].w - o.w : 0; // weight stride (floats)
        r.act = o.act;
        Run run;
        run.op = fused.size();
        run.table = b.reserve((int)(j - i) * 4);
        run.layers.assign(wr.ops.begin() + (long)i, wr.ops.begin() + (long)j);
        runs.push_back(std::move(run));
        fused.push_back(r);
        i = j;
      }
      wr.ops = std::move(fused);
      for (const auto& o : wr.ops)
      {
        wr.has_layers = wr.has_layers || o.type == WR_LAYER;
        wr.has_runs = wr.has_runs || o.type == WR_RUN;
        wr.has_rt_layers = wr.has_rt_layers || (o.type == WR_LAYER && policy != WrBuilder::JIT && !wr_layer_shape_is_exact(o.shape));
      }
    }
    wr.tab_rows = b.table(b.rows);
    wr.n_rows = (int)b.rows.size();
    wr.tab_pf = b.table(b.pf);
    wr.n_pf = (int)b.pf.size();
    wr.tab_ring = b.reserve((int)b.ring_of_slot.size());
    if (!b.ring_of_slot.empty())
      std::memcpy(&wr.blob[(size_t)wr.tab_ring], b.ring_of_slot.data(), b.ring_of_slot.size() * sizeof(int32_t));
    wr.tab_ops = b.reserve((int)wr.ops.size() * 16); // the macro-ops themselves: fetched from LDS, one op ahead
    const int hist_base = (int)wr.blob.size(); // LDS: weights, tables, program | rings
    for (auto& op : wr.ops)
      if (op.type == WR_LAYER || op.type == WR_ARRAY_END_K || (op.type == WR_POST_HEAD && op.ring > 0))
        op.hist += hist_base;
    for (const auto& run : runs)
    {
      wr.ops[run.op].hist = run.table;
      wr.ops[run.op].pad[0] = (int32_t)wr.run_recs.size(); // (the program compiled in: first record of this run)
      for (size_t l = 0; l < run.layers.size(); l++)
      {
        const WrOp& o = run.layers[l];
        const int32_t rec[4] = {o.w, o.hist + hist_base, o.ring, o.dil | (o.slot << 24)};
        std::memcpy(&wr.blob[(size_t)run.table + 4 * l], rec, sizeof(rec));
        wr.run_recs.push_back({rec[0], rec[1], rec[2], rec[3]});
      }
    }
    std::memcpy(&wr.blob[(size_t)wr.tab_ops], wr.ops.data(), wr.ops.size() * sizeof(WrOp));
    wr_program_cuts(wr.ops, wr.tab_rows, wr.split_op);
    wr.hist_floats = b.hist;
    wr.state_floats = (kWrPosInts + b.hist + 63) / 64 * 64;
    wr.lds_bytes = (hist_base + wr.hist_floats) * 4;
    if (wr.lds_bytes > kWrMaxLdsBytes)
      throw WrBuilder::Unsupported("more than 156 KB of weights and rings");
    wr.ok = true;
  }
}

// nam_wn_reg_kernel's plan: with the fully described ahead-of-time shapes if the model consists of them (the shipped
// examples: nothing to compile); else, when the caller offers a shape set, with the model's own shapes (the kernel is
// then compiled for them: wr_jit.cpp); else with the run-time-flag instantiations; else not at all (`why` says why).
void build_wr(const WaveNetSpec& wn, Plan& plan, WrShapeSet* jit_shapes)
{
  WrPlan wr;
  std::string why;
  bool done = false;
  auto attempt = [&](WrBuilder::Policy policy, WrShapeSet* dyn) {
    if (done)
      return;
    try
    {
      WrPlan w;
      build_wr_with(wn, w, policy, dyn);
      wr = std::move(w);
      done = true;
    }
    catch (const WrBuilder::Unsupported& e)
    {
      if (why.empty() || policy == WrBuilder::JIT)
        why = e.what();
    }
  };
  // Round 6: with a shape set on offer the per-model build comes FIRST — it compiles the plan's program in (every op a
  // constant expression: WrShapeSet::programs), which beats the ahead-of-time kernels walking the same program as data even
  // where they hold every shape (configs 4 and 5 of the bench: profiles/r06). NAM_HIP_WR_PROGRAM=0: round 5's order.
  static const bool program_first = [] { const char* e = std::getenv("NAM_HIP_WR_PROGRAM"); return !(e && e[0] == '0'); }();
  auto attempt_jit = [&]() {
    if (done || !jit_shapes)
      return;
    WrShapeSet trial = *jit_shapes; // (only a plan that succeeds leaves its shapes in the caller's set)
    attempt(WrBuilder::JIT, &trial);
    if (done)
    {
      WrShapeSet::Program pr;
      // the program as the code object holds it, twice: as it is (one wavefront per stream), and — a WR_RUN is ONE op to the
      // walked program (one dispatch for ten layers) and so could not be cut across the wavefronts of a two- / four-stage
      // launch — with every run cut at the quartile points of the program's work that fall inside it (sub-runs: their layers'
      // weights and ring records are consecutive; a sub-run costs one more exposed weight fetch, which is why the one-wavefront
      // form keeps the whole run) and the cuts taken again
      pr.ops = wr.ops;
      {
        // cost of every op as wr_program_cuts counts it (weights + 8), a run's layer by layer; the quartile points of the total
        std::vector<WrOp> probe = wr.ops;
        long total = 0;
        std::vector<long> cost(wr.ops.size(), 8);
        {
          std::vector<int> ws;
          auto weighs = [](const WrOp& op) {
            return op.type == WR_LAYER || op.type == WR_RUN || op.type == WR_ARRAY_BEGIN || op.type == WR_ARRAY_END || op.type == WR_ARRAY_END_K
                   || op.type == WR_POST_HEAD;
          };
          for (const auto& op : wr.ops)
            if (weighs(op))
              ws.push_back(op.w);
          ws.push_back(wr.tab_rows);
          std::sort(ws.begin(), ws.end());
          for (size_t i = 0; i < wr.ops.size(); i++)
          {
            if (weighs(wr.ops[i]))
            {
              const auto nx = std::upper_bound(ws.begin(), ws.end(), wr.ops[i].w);
              cost[i] += nx != ws.end() ? *nx - wr.ops[i].w : 0;
            }
            total += cost[i];
          }
        }
        // atoms: every op, a run layer by layer; the atom boundary closest to each quartile point of the total
        struct Atom
        {
          size_t op;
          int layer; // -1: not a run
          long cost;
        };
        std::vector<Atom> atoms;
        for (size_t i = 0; i < wr.ops.size(); i++)
        {
          const WrOp& o = wr.ops[i];
          if (o.type == WR_RUN && o.n_in >= 2)
            for (int l = 0; l < o.n_in; l++)
              atoms.push_back({i, l, cost[i] / o.n_in});
          else
            atoms.push_back({i, -1, cost[i]});
        }
        std::vector<std::vector<int>> cuts(wr.ops.size()); // per run: the layers a sub-run starts at
        for (int q = 1; q <= 3; q++)
        {
          long acc = 0, best = -1;
          size_t best_at = 0;
          for (size_t k = 1; k < atoms.size(); k++)
          {
            acc += atoms[k - 1].cost;
            const long d = std::labs(4 * acc - (long)q * total);
            if (best < 0 || d < best)
            {
              best = d;
              best_at = k;
            }
          }
          if (best_at > 0 && atoms[best_at].layer > 0) // the boundary lies inside a run: in front of this layer
            cuts[atoms[best_at].op].push_back(atoms[best_at].layer);
        }
        for (size_t i = 0; i < wr.ops.size(); i++)
        {
          const WrOp& o = wr.ops[i];
          if (o.type != WR_RUN || o.n_in < 2 || cuts[i].empty())
          {
            pr.ops_cut.push_back(o);
            continue;
          }
          std::vector<int> at = cuts[i];
          at.push_back(0);
          at.push_back(o.n_in);
          std::sort(at.begin(), at.end());
          at.erase(std::unique(at.begin(), at.end()), at.end());
          for (size_t k = 0; k + 1 < at.size(); k++)
          {
            WrOp sub = o;
            sub.w = o.w + at[k] * o.n_out; // (n_out: the layers' weight stride)
            sub.n_in = at[k + 1] - at[k];
            sub.pad[0] = o.pad[0] + at[k];
            pr.ops_cut.push_back(sub);
          }
        }
      }
      wr_program_cuts(pr.ops_cut, wr.tab_rows, pr.split_op);
      if (const char* e = std::getenv("NAM_HIP_WR_CUT2")) // (developer switch: the two-wave cut at this op, for A/B runs of the cost model)
        pr.split_op[3] = std::min(std::max(std::atoi(e), 1), (int)pr.ops_cut.size() - 1);
      pr.first_rec = (int)trial.run_recs.size();
      trial.run_recs.insert(trial.run_recs.end(), wr.run_recs.begin(), wr.run_recs.end());
      wr.program = (int)trial.programs.size();
      trial.programs.push_back(std::move(pr));
      *jit_shapes = std::move(trial);
      wr.jit = true;
    }
  };
  if (program_first)
    attempt_jit();
  attempt(WrBuilder::AOT_EXACT_ONLY, nullptr);
  attempt_jit();
  attempt(WrBuilder::AOT_ANY, nullptr);
  if (!done)
  {
    wr = WrPlan{};
    wr.why = why;
  }
  plan.wr = std::move(wr);
  if (plan.wr.ok)
    plan.state_floats = std::max(plan.state_floats, plan.wr.state_floats);
}

Plan build_wavenet_plan(const WaveNetSpec& wn, WrShapeSet* jit_shapes)
{
  validate_wavenet_geometry(wn);
  Plan plan;
  plan.arch = ARCH_WAVENET;
  plan.in_channels = wn.in_channels;
  plan.out_channels = wn.out_channels();
  plan.prewarm_samples = wn.prewarm_samples();
  Builder b(plan);
  const int in_rows = b.rows.alloc(wn.in_channels);
  {
    NamOp& op = b.push(OP_LOAD_IN);
    op.dst = in_rows;
    op.cout = wn.in_channels;
  }
  const int out_rows = b.wavenet(wn, in_rows);
  {
    NamOp& op = b.push(OP_STORE_OUT);
    op.src = out_rows;
    op.cin = plan.out_channels;
  }
  b.finish_stages(1); // history staging ops right behind OP_LOAD_IN
  b.push(OP_END);
  b.push(OP_END); // the interpreter reads one descriptor ahead
  plan.lds_rows = b.rows.high + 4; // + 4 spare rows: the kernel's four-row reads may run past the last tensor
  while (plan.blob.size() % 4)
    plan.blob.push_back(0.0f);
  plan.generic_blob_floats = (int)plan.blob.size();
  // per-stream state: [write positions: n_rings ints, padded to 64 words][rings...]
  const int table = (plan.n_rings + kBlock - 1) / kBlock * kBlock;
  for (auto& op : plan.ops)
    if ((op.type == OP_CONV || op.type == OP_STAGE) && op.state >= 0)
      op.state += table;
  plan.state_floats = (table + b.state_floats + kBlock - 1) / kBlock * kBlock;
  if (plan.state_floats == 0)
    plan.state_floats = kBlock;
  {
    // matrix-core kernel through zero-padded channels (lite: 12 -> 6 becomes 12 -> 8) when that makes it eligible;
    // the A1 kernels then share the padded ring layout, the generic kernel keeps the model's own
    WaveNetSpec padded;
    const size_t blob_mark = plan.blob.size();
    bool use_padded = false;
    if (pad_channels_for_mfma(wn, padded))
    {
      build_a1(padded, plan);
      use_padded = plan.a1.valid && plan.a1.ws_ok;
      if (!use_padded)
      {
        plan.blob.resize(blob_mark);
        plan.a1 = A1Plan{};
      }
    }
    if (!use_padded)
      build_a1(wn, plan);
    else
    {
      // the padded rings are longer rows: make sure the per-stream state covers them
      int need = 0;
      for (int a = 0; a < plan.a1.n_arrays; a++)
        for (int l = 0; l < plan.a1.arr[a].n_layers; l++)
          if (plan.a1.arr[a].ring_id[l] >= 0)
            need = std::max(need, plan.a1.arr[a].ring_off[l] + plan.a1.arr[a].channels * plan.a1.arr[a].ring_len[l]);
      plan.state_floats = std::max(plan.state_floats, (table + need + kBlock - 1) / kBlock * kBlock);
      plan.a1_padded_layout = true;
    }
  }
  if (plan.a1.valid)
  {
    for (int a = 0; a < plan.a1.n_arrays; a++)
      for (int l = 0; l < plan.a1.arr[a].n_layers; l++)
        if (plan.a1.arr[a].ring_id[l] >= 0)
          plan.a1.arr[a].ring_off[l] += table;
    for (int a = 0; a < plan.a1.n_arrays; a++)
      if (plan.a1.arr[a].head_ring_id >= 0)
        plan.a1.arr[a].head_ring_off += table;
    for (int j = 0; j < plan.a1.ws_jobs; j++)
    {
      // (an idle job has ring_b == 0 and never appends; its prefetch geometry points at the table, harmless)
      if (plan.a1.vdesc[j].flags & MV_RING)
        plan.a1.vdesc[j].ring_b += table * 4;
      plan.a1.vdesc[j].f_rbase += table * 4;
    }
    build_a1_kt(plan);
    build_a1_kp(plan);
    build_a1_il(plan);
    build_a1_q(plan);
  }
  build_wr(wn, plan, jit_shapes);
  return plan;
}

static Plan build_lstm_plan(const ModelSpec& model)
{
  const LSTMSpec& c = model.lstm;
  Plan plan;
  plan.arch = ARCH_LSTM;
  plan.in_channels = c.in_channels;
  plan.out_channels = c.out_channels;
  plan.prewarm_samples = model.prewarm_samples();
  LSTMPlan& L = plan.lstm;
  if (c.num_layers > 16)
    throw std::runtime_error("plan: LSTM with more than 16 layers is not supported on the device path");
  if (c.num_layers < 1)
    throw std::runtime_error("plan: LSTM with zero layers is not supported on the device path");
  if (c.in_channels != c.input_size)
    throw std::runtime_error("plan: LSTM in_channels must equal input_size");
  L.n_layers = c.num_layers;
  L.input_size = c.input_size;
  L.hidden = c.hidden_size;
  L.in_ch = c.in_channels;
  L.out_ch = c.out_channels;
  L.fast = model.fast_tanh ? 1 : 0;
  const float* w = c.weights.data();
  const int H = c.hidden_size;
  for (int l = 0; l < c.num_layers; l++)
  {
    const int I = l == 0 ? c.input_size : H;
    while (plan.blob.size() % 16)
      plan.blob.push_back(0.0f);
    L.layer_w[l] = (int)plan.blob.size();
    plan.blob.insert(plan.blob.end(), w, w + (size_t)4 * H * (I + H));
    w += (size_t)4 * H * (I + H);
    L.layer_b[l] = (int)plan.blob.size();
    plan.blob.insert(plan.blob.end(), w, w + 4 * H);
    w += 4 * H;
    // h0 then c0 (lstm.cpp:24-28)
    L.init_state.insert(L.init_state.end(), w, w + 2 * H);
    w += 2 * H;
  }
  L.head_w = (int)plan.blob.size();
  plan.blob.insert(plan.blob.end(), w, w + (size_t)c.out_channels * H);
  w += (size_t)c.out_channels * H;
  L.head_b = (int)plan.blob.size();
  plan.blob.insert(plan.blob.end(), w, w + c.out_channels);
  w += c.out_channels;
  if (w != c.weights.data() + c.weights.size())
    throw std::runtime_error("plan: LSTM weight stream not fully consumed");
  L.valid = 1;
  plan.state_floats = (c.num_layers * 2 * H + kBlock - 1) / kBlock * kBlock;

  // ---- MFMA tiles (plan.h) ----
  if (c.out_channels <= 16)
  {
    const int NT = (H + 3) / 4;
    while (plan.blob.size() % 64)
      plan.blob.push_back(0.0f);
    L.mf_off = (int)plan.blob.size();
    L.mf_nt = NT;
    std::vector<float> R;
    for (int l = 0; l < c.num_layers; l++)
    {
      const int I = l == 0 ? c.input_size : H;
      const int KI = (I + 3) / 4;
      const float* W = plan.blob.data() + L.layer_w[l]; // [4H][I + H] row-major (lstm.cpp:9-29)
      const float* B = plan.blob.data() + L.layer_b[l];
      L.mf_layer_tiles[l] = (int)R.size();
      for (int T = 0; T < NT; T++)
        for (int s = 0; s < KI + NT; s++)
          for (int lane = 0; lane < 64; lane++)
          {
            const int k = lane >> 4, i = lane & 15;
            const int unit = 4 * T + (i >> 2), gate = i & 3;
            const int e = 4 * (s < KI ? s : s - KI) + k; // input element this lane group feeds in this k-step
            const int col = s < KI ? (e < I ? e : -1) : (e < H ? I + e : -1);
            R.push_back((unit < H && col >= 0) ? W[(size_t)(gate * H + unit) * (I + H) + col] : 0.0f);
          }
      L.mf_layer_bias[l] = (int)R.size();
      for (int T = 0; T < NT; T++)
        for (int u = 0; u < 4; u++)
          for (int gate = 0; gate < 4; gate++)
            R.push_back(4 * T + u < H ? B[gate * H + 4 * T + u] : 0.0f);
    }
    const float* Wh = plan.blob.data() + L.head_w; // [out][H]
    const float* Bh = plan.blob.data() + L.head_b;
    L.mf_head_tiles = (int)R.size();
    for (int s = 0; s < NT; s++)
      for (int lane = 0; lane < 64; lane++)
      {
        const int k = lane >> 4, o = lane & 15, e = 4 * s + k;
        R.push_back((o < c.out_channels && e < H) ? Wh[(size_t)o * H + e] : 0.0f);
      }
    L.mf_head_bias = (int)R.size();
    for (int o = 0; o < 16; o++)
      R.push_back(o < c.out_channels ? Bh[o] : 0.0f);
    while (R.size() % 64)
      R.push_back(0.0f);
    L.mf_floats = (int)R.size();
    plan.blob.insert(plan.blob.end(), R.begin(), R.end());
    // LDS: region | h [2][layers][4 NT][16] | c [layers][4 NT][16] | in [in_ch][16][65] | out [out_ch][16][65]
    const long lds_floats = (long)L.mf_floats + 3L * c.num_layers * 4 * NT * 16 + (long)(c.in_channels + c.out_channels) * 16 * 65;
    L.mf_lds_bytes = (int)(lds_floats * 4);
    L.mf_ok = lds_floats * 4 <= 150 * 1024 ? 1 : 0;
  }
  return plan;
}

Plan build_plan(const ModelSpec& model, WrShapeSet* jit_shapes)
{
  if (model.arch == ARCH_WAVENET)
  {
    if (model.wavenet.slimmable)
      return build_wavenet_plan(slim_wavenet(model.wavenet, channels_for_ratio(model.wavenet, 1.0)), jit_shapes);
    return build_wavenet_plan(model.wavenet, jit_shapes);
  }
  if (model.arch == ARCH_LSTM)
    return build_lstm_plan(model);
  throw std::runtime_error("plan: unknown architecture");
}

std::string Plan::describe() const
{
  std::stringstream ss;
  ss << "arch=" << arch << " in=" << in_channels << " out=" << out_channels << " ops=" << ops.size()
     << " blob=" << blob.size() << " lds_rows=" << lds_rows << " rings=" << n_rings
     << " state_floats=" << state_floats << " a1=" << a1.valid << " lstm=" << lstm.valid
     << " prewarm=" << prewarm_samples;
  return ss.str();
}

} // namespace namhip
