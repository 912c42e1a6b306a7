// plan_wr.cpp — nam_wn_reg_kernel's plan: the macro-op program, padded dense / matrix-form weights, the shape sets and the
// header of the per-model compile, the cuts of the two- / four-wave launches. See plan_internal.h.
#include "plan_internal.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <sstream>

namespace namhip
{
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

} // namespace namhip
