// plan_ops.cpp — the op program (the interpreter's: nam_generic_kernel runs every feature). A straight transcription of the
// reference's per-block control flow:
//   WaveNet::process            NAM/wavenet/model.cpp:822-910
//   LayerArray::Process(Inner)  NAM/wavenet/model.cpp:463-549
//   Layer::Process              NAM/wavenet/model.cpp:183-393
//   detail::Head::process       NAM/wavenet/model.cpp:86-103
// the weight blob built by walking the flat weight stream in set_weights_ order (model.cpp:152-181, 563-569, 661-683; Conv1D
// conv1d.cpp:40-55; Conv1x1 dsp.cpp:384-397). See plan_internal.h.
#include "plan_internal.h"

#include <algorithm>
#include <cstdlib>
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

} // namespace

// The op program of a WaveNet (the interpreter's: every feature) + the per-stream state layout it implies. Returns the size of
// the write-position table in front of the rings (floats).
int build_op_program(const WaveNetSpec& wn, Plan& plan)
{
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
  return table;
}

} // namespace namhip
