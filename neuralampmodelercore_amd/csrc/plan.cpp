// plan.cpp — ModelSpec -> device Plan (host side): the entry points. The pieces: plan_ops.cpp (the op program), plan_a1.cpp (the
// A1-family kernels' plans), plan_wr.cpp (nam_wn_reg_kernel's plan), this file (a WaveNet's plan put together, the LSTM plan).
#include "plan_internal.h"
#include <cstdlib>
#include "kp_table.h"
#include "aq_table.h"

#include <algorithm>
#include <cstring>
#include <sstream>

namespace namhip
{
Plan build_wavenet_plan(const WaveNetSpec& wn, WrShapeSet* jit_shapes)
{
  validate_wavenet_geometry(wn);
  Plan plan;
  plan.arch = ARCH_WAVENET;
  plan.in_channels = wn.in_channels;
  plan.out_channels = wn.out_channels();
  plan.prewarm_samples = wn.prewarm_samples();
  const int table = build_op_program(wn, plan);
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

