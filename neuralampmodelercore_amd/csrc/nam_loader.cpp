// nam_loader.cpp — .nam JSON -> ModelSpec (host side, no GPU).
//
// Behavioural mirror of the reference's load path, written against its schema:
//   validate_nam_file            NAM/nam_file.cpp:9-40
//   verify_config_version        NAM/get_dsp.cpp:18-39,113-128 ; limits NAM/get_dsp.h:66-67
//   populate_dsp_data / metadata NAM/get_dsp.cpp:141-154,229-259,275-281
//   WaveNet parse_config_json    NAM/wavenet/model.cpp:913-1276 (all legacy forms)
//   ActivationConfig::from_json  NAM/activations.cpp:59-130
//   Layer ctor validation        NAM/wavenet/detail.h:54-85
//   WaveNet ctor validation      NAM/wavenet/model.cpp:600-650 ; weight count :661-683
//   LSTM parse_config_json       NAM/lstm.cpp:170-181
//   SlimmableWavenet             NAM/wavenet/slimmable.cpp:80-294,352-395,541-585
#include "model_spec.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>

#include "json_min.h"

namespace namhip
{
using json::Value;

// ---------------------------------------------------------------------------------------------
// Version gate
// ---------------------------------------------------------------------------------------------
namespace
{
struct Ver
{
  int major = 0, minor = 0, patch = 0;
  bool operator<(const Ver& o) const
  {
    if (major != o.major)
      return major < o.major;
    if (minor != o.minor)
      return minor < o.minor;
    return patch < o.patch;
  }
};

bool parse_semver(const std::string& s, Ver& v)
{
  // ^\d+\.\d+\.\d+$
  int parts[3] = {0, 0, 0};
  int idx = 0;
  bool have_digit = false;
  for (size_t i = 0; i < s.size(); i++)
  {
    const char c = s[i];
    if (c >= '0' && c <= '9')
    {
      if (parts[idx] > 100000000)
        return false;
      parts[idx] = parts[idx] * 10 + (c - '0');
      have_digit = true;
    }
    else if (c == '.')
    {
      if (!have_digit || idx == 2)
        return false;
      idx++;
      have_digit = false;
    }
    else
      return false;
  }
  if (idx != 2 || !have_digit)
    return false;
  v.major = parts[0];
  v.minor = parts[1];
  v.patch = parts[2];
  return true;
}
} // namespace

int version_support(const std::string& version)
{
  Ver parsed, latest, earliest;
  if (!parse_semver(version, parsed))
    return 0;
  parse_semver("0.7.0", latest); // LATEST_FULLY_SUPPORTED_NAM_FILE_VERSION
  parse_semver("0.5.0", earliest); // EARLIEST_SUPPORTED_NAM_FILE_VERSION
  if (parsed < earliest)
    return 0;
  if (parsed.major > latest.major || parsed.minor > latest.minor)
    return 0;
  if (latest < parsed)
    return 1;
  return 2;
}

static void verify_config_version(const std::string& version)
{
  const int s = version_support(version);
  if (s == 0)
    throw std::runtime_error("Model config is an unsupported version " + version + ".");
  if (s == 1)
    std::cerr << "Model config is a partially-supported version " << version << ". Continuing with partial support."
              << std::endl;
}

// ---------------------------------------------------------------------------------------------
// Activations
// ---------------------------------------------------------------------------------------------
static int act_type_from_name(const std::string& name)
{
  static const std::pair<const char*, int> table[] = {
    {"Tanh", ACT_TANH},           {"Hardtanh", ACT_HARDTANH},   {"Fasttanh", ACT_FASTTANH},
    {"ReLU", ACT_RELU},           {"LeakyReLU", ACT_LEAKYRELU}, {"PReLU", ACT_PRELU},
    {"Sigmoid", ACT_SIGMOID},     {"SiLU", ACT_SILU},           {"Hardswish", ACT_HARDSWISH},
    {"LeakyHardtanh", ACT_LEAKYHARDTANH}, {"LeakyHardTanh", ACT_LEAKYHARDTANH}, {"Softsign", ACT_SOFTSIGN}};
  for (const auto& e : table)
    if (name == e.first)
      return e.second;
  throw std::runtime_error("Unknown activation type: " + name);
}

// FastLUTActivation's constructor (activations.h:374-388): the table is f(min + i * step) in float arithmetic, built
// with the host's libm exactly as the reference builds it; the device only interpolates.
static ActSpec make_lut_activation(const LutSpec& l)
{
  ActSpec a;
  a.type = ACT_LUT;
  const float step = (l.max_x - l.min_x) / (float)(l.n_points - 1);
  a.p[0] = l.min_x;
  a.p[1] = l.max_x;
  a.p[2] = 1.0f / step;
  a.p[3] = (float)l.n_points;
  a.slopes.reserve((size_t)l.n_points);
  for (int i = 0; i < l.n_points; i++)
  {
    const float x = l.min_x + (float)i * step;
    float y;
    if (l.act_type == ACT_TANH)
      y = std::tanh(x);
    else
    {
      const float sg = 1.0f / (1.0f + expf(-x)); // activations.h:71-74
      y = l.act_type == ACT_SIGMOID ? sg : x * sg; // swish, activations.h:107-110
    }
    a.slopes.push_back(y);
  }
  return a;
}

// ActivationConfig::from_json + Activation::get_activation(config) folded together.
static ActSpec parse_activation(const Value& j, const LoadOptions& lo)
{
  ActSpec a;
  if (j.is_string())
  {
    a.type = act_type_from_name(j.str);
    // defaults of the registry singletons (activations.cpp:3-13) / get_activation fallbacks (:132-166)
    if (a.type == ACT_LEAKYRELU)
      a.p[0] = 0.01f;
    else if (a.type == ACT_PRELU)
      a.slopes = {0.01f};
    else if (a.type == ACT_LEAKYHARDTANH)
    {
      a.p[0] = -1.0f;
      a.p[1] = 1.0f;
      a.p[2] = 0.01f;
      a.p[3] = 0.01f;
    }
  }
  else if (j.is_object())
  {
    a.type = act_type_from_name(j.at("type").as_string());
    if (a.type == ACT_PRELU)
    {
      if (j.contains("negative_slope"))
        a.slopes = {(float)j.at("negative_slope").as_double()};
      else if (j.contains("negative_slopes"))
      {
        for (const auto& s : j.at("negative_slopes").arr)
          a.slopes.push_back((float)s.as_double());
      }
      else
        a.slopes = {0.01f};
      if (a.slopes.empty())
        a.slopes = {0.01f};
    }
    else if (a.type == ACT_LEAKYRELU)
      a.p[0] = (float)j.value_double("negative_slope", 0.01f);
    else if (a.type == ACT_LEAKYHARDTANH)
    {
      a.p[0] = (float)j.value_double("min_val", -1.0f);
      a.p[1] = (float)j.value_double("max_val", 1.0f);
      a.p[2] = (float)j.value_double("min_slope", 0.01f);
      a.p[3] = (float)j.value_double("max_slope", 0.01f);
    }
  }
  else
    throw std::runtime_error("Invalid activation config: expected string or object");
  // Activation::enable_lut replaces the registry entry of "Tanh" / "Sigmoid" / "SiLU" with a FastLUTActivation
  // (activations.cpp:189-212, activations.h:371-422); a model binds whatever the registry holds when it is built.
  // It wins over fast tanh (same as calling enable_fast_tanh() and then enable_lut("Tanh", ...)).
  for (const LutSpec& l : lo.luts)
    if (l.act_type == a.type)
      return make_lut_activation(l);
  // Activation::enable_fast_tanh swaps only the "Tanh" entry (activations.cpp:168-177)
  if (lo.fast_tanh && a.type == ACT_TANH)
    a.type = ACT_FASTTANH;
  return a;
}

static ActSpec simple_act(int type)
{
  ActSpec a;
  a.type = type;
  return a;
}

// ---------------------------------------------------------------------------------------------
// WaveNet
// ---------------------------------------------------------------------------------------------
static int parse_gating_mode(const std::string& s)
{
  if (s == "gated")
    return GATING_GATED;
  if (s == "blended")
    return GATING_BLENDED;
  if (s == "none")
    return GATING_NONE;
  throw std::runtime_error("Invalid gating_mode: " + s);
}

static FilmSpec parse_film(const Value& lc, const char* key)
{
  FilmSpec f;
  const Value* v = lc.find(key);
  if (!v || (v->is_bool() && !v->b))
    return f; // inactive
  f.active = v->value_bool("active", true);
  f.shift = v->value_bool("shift", true);
  f.groups = v->value_int("groups", 1);
  return f;
}

static void check_divisible(int in_ch, int out_ch, int groups)
{
  // Conv1D::set_size_ conv1d.cpp:61-71 / Conv1x1 ctor dsp.cpp:313-323
  if (groups <= 0 || in_ch % groups != 0)
    throw std::runtime_error("in_channels (" + std::to_string(in_ch) + ") must be divisible by numGroups ("
                             + std::to_string(groups) + ")");
  if (out_ch % groups != 0)
    throw std::runtime_error("out_channels (" + std::to_string(out_ch) + ") must be divisible by numGroups ("
                             + std::to_string(groups) + ")");
}

static LayerArraySpec parse_layer_array(const Value& lc, size_t i, const LoadOptions& lo)
{
  LayerArraySpec p;
  const std::string la = "Layer array " + std::to_string(i);
  p.groups_input = lc.value_int("groups_input", 1);
  p.groups_input_mixin = lc.value_int("groups_input_mixin", 1);
  p.channels = lc.at("channels").as_int();
  p.bottleneck = lc.value_int("bottleneck", p.channels);
  if (const Value* l1 = lc.find("layer1x1"))
  {
    p.layer1x1_active = l1->at("active").as_bool();
    p.layer1x1_groups = l1->at("groups").as_int();
  }
  p.input_size = lc.at("input_size").as_int();
  p.condition_size = lc.at("condition_size").as_int();

  const Value* head = lc.find("head");
  if (head && !head->is_null())
  {
    if (!head->is_object())
      throw std::runtime_error(la + ": 'head' must be a JSON object");
    p.head_size = head->at("out_channels").as_int();
    if (head->contains("head_dilation"))
      p.head_dilation = head->at("head_dilation").as_int();
    p.head_kernel_size = head->at("kernel_size").as_int();
    p.head_bias = head->at("bias").as_bool();
  }
  else if (lc.contains("head_size"))
  {
    p.head_size = lc.at("head_size").as_int();
    p.head_kernel_size = 1;
    p.head_bias = lc.at("head_bias").as_bool();
  }
  else
    throw std::runtime_error(la
                             + ": expected 'head' object with out_channels, kernel_size, and bias, "
                               "or legacy 'head_size' and 'head_bias'");
  if (p.head_kernel_size < 1)
    throw std::runtime_error(la + ": head.kernel_size must be >= 1");

  for (const auto& d : lc.at("dilations").arr)
    p.dilations.push_back(d.as_int());
  const size_t n = p.dilations.size();

  const bool has_ks = lc.contains("kernel_size"), has_kss = lc.contains("kernel_sizes");
  if (has_ks && has_kss)
    throw std::runtime_error(la + ": only one of kernel_size (int) or kernel_sizes (array) may be provided");
  if (has_kss)
  {
    const Value& ks = lc.at("kernel_sizes");
    if (!ks.is_array())
      throw std::runtime_error(la + ": kernel_sizes must be an array");
    for (const auto& k : ks.arr)
      p.kernel_sizes.push_back(k.as_int());
    if (p.kernel_sizes.size() != n)
      throw std::runtime_error(la + ": kernel_sizes array size (" + std::to_string(p.kernel_sizes.size())
                               + ") must match dilations size (" + std::to_string(n) + ")");
  }
  else if (has_ks)
    p.kernel_sizes.assign(n, lc.at("kernel_size").as_int());
  else
    throw std::runtime_error(la + ": either kernel_size (int) or kernel_sizes (array) must be provided");

  const Value& act = lc.at("activation");
  if (act.is_array())
  {
    for (const auto& a : act.arr)
      p.activations.push_back(parse_activation(a, lo));
    if (p.activations.size() != n)
      throw std::runtime_error(la + ": activation array size (" + std::to_string(p.activations.size())
                               + ") must match dilations size (" + std::to_string(n) + ")");
  }
  else
    p.activations.assign(n, parse_activation(act, lo));

  const Value* sa = lc.find("secondary_activation");
  if (const Value* gm = lc.find("gating_mode"))
  {
    if (gm->is_array())
    {
      for (const auto& g : gm->arr)
      {
        const int mode = parse_gating_mode(g.as_string());
        p.gating_modes.push_back(mode);
        if (mode != GATING_NONE)
        {
          if (sa)
          {
            if (sa->is_array())
            {
              if (p.gating_modes.size() > sa->arr.size())
                throw std::runtime_error(la + ": secondary_activation array size must be at least "
                                         + std::to_string(p.gating_modes.size()));
              p.secondary_activations.push_back(parse_activation(sa->arr[p.gating_modes.size() - 1], lo));
            }
            else
              p.secondary_activations.push_back(parse_activation(*sa, lo));
          }
          else
            p.secondary_activations.push_back(simple_act(ACT_SIGMOID));
        }
        else
          p.secondary_activations.push_back(ActSpec{});
      }
      if (p.gating_modes.size() != n)
        throw std::runtime_error(la + ": gating_mode array size (" + std::to_string(p.gating_modes.size())
                                 + ") must match dilations size (" + std::to_string(n) + ")");
      if (sa && sa->is_array() && sa->arr.size() != n)
        throw std::runtime_error(la + ": secondary_activation array size (" + std::to_string(sa->arr.size())
                                 + ") must match dilations size (" + std::to_string(n) + ")");
    }
    else
    {
      const int mode = parse_gating_mode(gm->as_string());
      p.gating_modes.assign(n, mode);
      ActSpec s;
      if (mode != GATING_NONE)
        s = sa ? parse_activation(*sa, lo) : simple_act(ACT_SIGMOID);
      p.secondary_activations.assign(n, s);
    }
  }
  else if (const Value* g = lc.find("gated"))
  {
    const bool gated = g->as_bool();
    p.gating_modes.assign(n, gated ? GATING_GATED : GATING_NONE);
    p.secondary_activations.assign(n, gated ? simple_act(ACT_SIGMOID) : ActSpec{});
  }
  else
  {
    p.gating_modes.assign(n, GATING_NONE);
    p.secondary_activations.assign(n, ActSpec{});
  }

  p.head1x1_out = p.channels;
  if (const Value* h1 = lc.find("head1x1"))
  {
    p.head1x1_active = h1->at("active").as_bool();
    p.head1x1_out = h1->at("out_channels").as_int();
    p.head1x1_groups = h1->at("groups").as_int();
  }

  static const char* film_keys[FILM_COUNT] = {"conv_pre_film",        "conv_post_film",      "input_mixin_pre_film",
                                              "input_mixin_post_film", "activation_pre_film", "activation_post_film",
                                              "layer1x1_post_film",    "head1x1_post_film"};
  for (int k = 0; k < FILM_COUNT; k++)
    p.film[k] = parse_film(lc, film_keys[k]);

  if (p.film[FILM_LAYER1X1_POST].active && !p.layer1x1_active)
    throw std::runtime_error(la + ": layer1x1_post_film cannot be active when layer1x1.active is false");

  // Layer ctor validation (detail.h:58-85)
  if (!p.layer1x1_active && p.bottleneck != p.channels)
    throw std::invalid_argument("When layer1x1.active is false, bottleneck (" + std::to_string(p.bottleneck)
                                + ") must equal channels (" + std::to_string(p.channels) + ")");
  if (p.film[FILM_HEAD1X1_POST].active && !p.head1x1_active)
    throw std::invalid_argument("Do not use post-head 1x1 FiLM if there is no head 1x1");

  // group divisibility, as the Conv1D / Conv1x1 constructors would throw
  bool any_gated = false;
  for (int m : p.gating_modes)
    any_gated |= (m != GATING_NONE);
  for (size_t l = 0; l < n; l++)
  {
    const int zc = p.gating_modes[l] != GATING_NONE ? 2 * p.bottleneck : p.bottleneck;
    check_divisible(p.channels, zc, p.groups_input);
    check_divisible(p.condition_size, zc, p.groups_input_mixin);
    const int dims[FILM_COUNT] = {p.channels, zc, p.condition_size, zc, zc, p.bottleneck, p.channels, p.head1x1_out};
    for (int k = 0; k < FILM_COUNT; k++)
      if (p.film[k].active)
        check_divisible(p.condition_size, (p.film[k].shift ? 2 : 1) * dims[k], p.film[k].groups);
  }
  if (p.layer1x1_active)
    check_divisible(p.bottleneck, p.channels, p.layer1x1_groups);
  if (p.head1x1_active)
    check_divisible(p.bottleneck, p.head1x1_out, p.head1x1_groups);
  (void)any_gated;
  return p;
}

static long conv_weights(int in_ch, int out_ch, int k, int groups, bool bias)
{
  return (long)k * in_ch * out_ch / groups + (bias ? out_ch : 0);
}

long WaveNetSpec::expected_weight_count() const
{
  long n = 0;
  for (const auto& p : arrays)
  {
    n += conv_weights(p.input_size, p.channels, 1, 1, false); // rechannel
    for (int l = 0; l < p.num_layers(); l++)
    {
      const int zc = p.gating_modes[l] != GATING_NONE ? 2 * p.bottleneck : p.bottleneck;
      n += conv_weights(p.channels, zc, p.kernel_sizes[l], p.groups_input, true);
      n += conv_weights(p.condition_size, zc, 1, p.groups_input_mixin, false);
      if (p.layer1x1_active)
        n += conv_weights(p.bottleneck, p.channels, 1, p.layer1x1_groups, true);
      if (p.head1x1_active)
        n += conv_weights(p.bottleneck, p.head1x1_out, 1, p.head1x1_groups, true);
      const int dims[FILM_COUNT] = {p.channels, zc, p.condition_size, zc, zc, p.bottleneck, p.channels, p.head1x1_out};
      for (int k = 0; k < FILM_COUNT; k++)
      {
        bool active = p.film[k].active;
        if (k == FILM_LAYER1X1_POST && !p.layer1x1_active)
          active = false;
        if (k == FILM_HEAD1X1_POST && !p.head1x1_active)
          active = false;
        if (active)
          n += conv_weights(p.condition_size, (p.film[k].shift ? 2 : 1) * dims[k], 1, p.film[k].groups, true);
      }
    }
    n += conv_weights(p.head_output_size(), p.head_size, p.head_kernel_size, 1, p.head_bias);
  }
  if (with_head)
  {
    int cin = head.in_channels;
    for (size_t i = 0; i < head.kernel_sizes.size(); i++)
    {
      const int cout = (i + 1 == head.kernel_sizes.size()) ? head.out_channels : head.channels;
      n += conv_weights(cin, cout, head.kernel_sizes[i], 1, true);
      cin = cout;
    }
  }
  return n + 1; // head_scale
}

int ModelSpec::container_index(double val) const
{
  for (size_t i = 0; i < sub_max_value.size(); i++)
    if (val < sub_max_value[i])
      return (int)i;
  return (int)sub_max_value.size() - 1;
}

int ModelSpec::prewarm_samples() const
{
  if (arch == ARCH_WAVENET)
    return wavenet.prewarm_samples();
  if (arch == ARCH_CONTAINER) // of the active submodel; a fresh container has the last one active (container.cpp:49,139-144)
    return submodels.back()->prewarm_samples();
  // LSTM::GetPrewarmSamples lstm.cpp:127-134
  const int r = (int)(0.5 * sample_rate);
  return r <= 0 ? 1 : r;
}

int WaveNetSpec::prewarm_samples() const
{
  int n = condition_dsp ? condition_dsp->prewarm_samples() : 1;
  for (const auto& p : arrays)
  {
    for (int l = 0; l < p.num_layers(); l++)
      n += p.dilations[l] * (p.kernel_sizes[l] - 1);
    n += p.head_dilation * (p.head_kernel_size - 1);
  }
  if (with_head)
  {
    int rf = 1;
    for (int k : head.kernel_sizes)
      rf += k - 1;
    n += rf - 1;
  }
  return n;
}

long LSTMSpec::expected_weight_count() const
{
  long n = 0;
  for (int i = 0; i < num_layers; i++)
  {
    const long I = i == 0 ? input_size : hidden_size;
    n += 4L * hidden_size * (I + hidden_size) + 4L * hidden_size + 2L * hidden_size;
  }
  return n + (long)out_channels * hidden_size + out_channels;
}

static std::shared_ptr<ModelSpec> build_model(const Value& root, const LoadOptions& lo);

static bool config_is_slimmable(const Value& config)
{
  // config_is_slimmable_wavenet model.cpp:1290-1308
  const Value* layers = config.find("layers");
  if (!layers || !layers->is_array())
    return false;
  for (const auto& lc : layers->arr)
  {
    const Value* sl = lc.find("slimmable");
    if (!sl || !sl->is_object())
      continue;
    const std::string method = sl->value_string("method", "");
    if (method != "slice_channels_uniform")
    {
      if (!method.empty())
        throw std::runtime_error("SlimmableWavenet: unsupported slimmable method '" + method + "'");
      continue;
    }
    return true;
  }
  return false;
}

static void parse_wavenet(const Value& config_in, double sample_rate, const LoadOptions& lo, ModelSpec& out)
{
  WaveNetSpec& wc = out.wavenet;
  const bool slimmable = config_is_slimmable(config_in);
  // SlimmableWavenetConfig::create: support wrapped {"model": {...}} (slimmable.cpp:544)
  const Value& config = (slimmable && config_in.contains("model")) ? config_in.at("model") : config_in;

  const Value* cd = config.find("condition_dsp");
  if (cd && !cd->is_null())
  {
    wc.condition_dsp = build_model(*cd, lo);
    if (wc.condition_dsp->sample_rate != sample_rate)
    {
      std::stringstream ss;
      ss << "Condition DSP expected sample rate (" << wc.condition_dsp->sample_rate
         << ") doesn't match WaveNet expected sample rate (" << sample_rate << "!\n";
      throw std::runtime_error(ss.str());
    }
  }
  const Value& layers = config.at("layers");
  for (size_t i = 0; i < layers.size(); i++)
    wc.arrays.push_back(parse_layer_array(layers.at(i), i, lo));

  const Value* hj = config.find("head");
  wc.with_head = hj && !hj->is_null();
  wc.head_scale_json = (float)config.at("head_scale").as_double();
  wc.in_channels = config.value_int("in_channels", 1);
  if (wc.arrays.empty())
    throw std::runtime_error("WaveNet config requires at least one layer array");
  if (wc.with_head)
  {
    const int implied_in = wc.arrays.back().head_size;
    const Value* ic = hj->find("in_channels");
    if (ic && !ic->is_null() && ic->as_int() != implied_in)
    {
      std::stringstream ss;
      ss << "WaveNet config: head.in_channels (" << ic->as_int() << ") must equal last layer's head_size ("
         << implied_in << ")";
      throw std::runtime_error(ss.str());
    }
    wc.head.in_channels = implied_in;
    wc.head.channels = hj->at("channels").as_int();
    wc.head.out_channels = hj->at("out_channels").as_int();
    for (const auto& k : hj->at("kernel_sizes").arr)
    {
      if (k.as_int() < 1)
        throw std::runtime_error("WaveNet Head: kernel_sizes entries must be >= 1");
      wc.head.kernel_sizes.push_back(k.as_int());
    }
    wc.head.activation = parse_activation(hj->at("activation"), lo);
    if (wc.head.kernel_sizes.empty())
      throw std::runtime_error("WaveNet config: head.kernel_sizes must be non-empty");
  }

  // WaveNet ctor checks (model.cpp:600-650)
  if (wc.in_channels <= 0)
    throw std::runtime_error("Channel counts must be positive");
  if (wc.condition_dsp)
  {
    if (wc.condition_dsp->in_channels() != wc.in_channels)
    {
      std::stringstream ss;
      ss << "input channels of WaveNet (" << wc.in_channels << ") don't match input channels of condition DSP ("
         << wc.condition_dsp->in_channels() << "!\n";
      throw std::runtime_error(ss.str());
    }
  }
  for (size_t i = 0; i < wc.arrays.size(); i++)
  {
    if (wc.condition_dsp && wc.arrays[i].condition_size != wc.condition_dsp->out_channels())
    {
      std::stringstream ss;
      ss << "condition_size of layer " << i << " (" << wc.arrays[i].condition_size
         << ") doesn't match output channels of condition DSP (" << wc.condition_dsp->out_channels() << "!\n";
      throw std::runtime_error(ss.str());
    }
    if (i > 0 && wc.arrays[i].channels != wc.arrays[i - 1].head_size)
    {
      std::stringstream ss;
      ss << "channels of layer " << i << " (" << wc.arrays[i].channels << ") doesn't match head_size of preceding layer ("
         << wc.arrays[i - 1].head_size << "!\n";
      throw std::runtime_error(ss.str());
    }
  }

  if (slimmable)
  {
    wc.slimmable = true;
    if (wc.with_head)
      throw std::runtime_error("SlimmableWavenet: post-stack head is not supported");
    bool any = false;
    for (size_t i = 0; i < layers.size(); i++)
    {
      std::vector<int> allowed;
      const Value* sl = layers.at(i).find("slimmable");
      if (sl && sl->is_object())
      {
        const std::string method = sl->value_string("method", "");
        if (method != "slice_channels_uniform")
          throw std::runtime_error("SlimmableWavenet: unsupported slimmable method '" + method + "'");
        const Value* kw = sl->find("kwargs");
        const Value* ac = kw ? kw->find("allowed_channels") : nullptr;
        if (ac)
          for (const auto& c : ac->arr)
            allowed.push_back(c.as_int());
        else
          for (int c = 1; c <= wc.arrays[i].channels; c++)
            allowed.push_back(c);
      }
      if (!allowed.empty())
      {
        any = true;
        for (size_t j = 1; j < allowed.size(); j++)
          if (allowed[j] <= allowed[j - 1])
            throw std::runtime_error("SlimmableWavenet: allowed_channels must be sorted ascending");
        if (allowed.back() != wc.arrays[i].channels)
          throw std::runtime_error(
            "SlimmableWavenet: last allowed_channels entry must equal the full channel count for that array");
      }
      wc.allowed_channels.push_back(std::move(allowed));
    }
    if (!any)
      throw std::runtime_error("SlimmableWavenet: at least one layer array must have allowed_channels");
  }
}

static void check_wavenet_weights(const WaveNetSpec& wc)
{
  const long expect = wc.expected_weight_count();
  const long got = (long)wc.weights.size();
  if (expect != got)
  {
    // model.cpp:671-682 — the reference reports either direction as "Weight mismatch"
    std::stringstream ss;
    if (got > expect)
    {
      // the reference reports the position of the first weight whose VALUE equals the first unassigned one
      // (model.cpp:674-679), which is `expect + 1` unless that value also occurs earlier in the stream
      long pos = expect;
      for (long i = 0; i < got; i++)
        if (wc.weights[(size_t)i] == wc.weights[(size_t)expect])
        {
          pos = i;
          break;
        }
      ss << "Weight mismatch: assigned " << pos + 1 << " weights, but " << got << " were provided.";
    }
    else
      ss << "Weight mismatch: provided " << got << " weights, but the model expects more.";
    throw std::runtime_error(ss.str());
  }
}

static std::shared_ptr<ModelSpec> build_model(const Value& root, const LoadOptions& lo_outer)
{
  // (the caller's own version gate covered this document only: nested documents — a condition_dsp, a container's
  // submodels — go through the built-in gate)
  LoadOptions lo = lo_outer;
  lo.skip_version_gate = false;
  auto m = std::make_shared<ModelSpec>();
  m->fast_tanh = lo.fast_tanh;
  m->version = root.at("version").as_string();
  if (!lo_outer.skip_version_gate)
    verify_config_version(m->version);
  const Value* w = root.find("weights");
  if (!w)
    throw std::runtime_error("Corrupted model file is missing weights.");
  std::vector<float> weights;
  weights.reserve(w->arr.size());
  for (const auto& x : w->arr)
    weights.push_back((float)x.as_double());
  const std::string arch = root.at("architecture").as_string();
  const Value& config = root.at("config");
  m->architecture_name = arch;
  m->config_text = json::dump(config);
  if (const Value* mdv = root.find("metadata"))
    m->metadata_text = json::dump(*mdv);
  m->sample_rate = root.contains("sample_rate") ? root.at("sample_rate").as_double() : -1.0;

  const Value* md = root.find("metadata");
  if (md && md->is_object())
  {
    auto extract = [&](const char* key, bool& has, double& val) {
      const Value* v = md->find(key);
      if (v && !v->is_null())
      {
        has = true;
        val = v->as_double();
      }
    };
    extract("loudness", m->has_loudness, m->loudness);
    extract("input_level_dbu", m->has_input_level, m->input_level);
    extract("output_level_dbu", m->has_output_level, m->output_level);
  }

  if (arch == "WaveNet")
  {
    m->arch = ARCH_WAVENET;
    parse_wavenet(config, m->sample_rate, lo, *m);
    m->wavenet.weights = std::move(weights);
    check_wavenet_weights(m->wavenet);
  }
  else if (arch == "LSTM")
  {
    m->arch = ARCH_LSTM;
    LSTMSpec& c = m->lstm;
    c.num_layers = config.at("num_layers").as_int();
    c.input_size = config.at("input_size").as_int();
    c.hidden_size = config.at("hidden_size").as_int();
    c.in_channels = config.value_int("in_channels", 1);
    c.out_channels = config.value_int("out_channels", 1);
    if (c.in_channels <= 0 || c.out_channels <= 0)
      throw std::runtime_error("Channel counts must be positive");
    c.weights = std::move(weights);
    if (c.expected_weight_count() != (long)c.weights.size())
      throw std::runtime_error("LSTM weight mismatch: model expects " + std::to_string(c.expected_weight_count())
                               + " weights, but " + std::to_string(c.weights.size()) + " were provided.");
  }
  else if (arch == "SlimmableContainer")
  {
    // ContainerConfig::create + ContainerModel ctor (container.cpp:17-50,150-171)
    m->arch = ARCH_CONTAINER;
    const Value* subs = config.find("submodels");
    if (!subs || !subs->is_array() || subs->arr.empty())
      throw std::runtime_error("SlimmableContainer: 'submodels' must be a non-empty array");
    for (const Value& entry : subs->arr)
    {
      m->sub_max_value.push_back(entry.at("max_value").as_double());
      m->submodels.push_back(build_model(entry.at("model"), lo)); // each one is a full .nam document
    }
    for (size_t i = 1; i < m->sub_max_value.size(); i++)
      if (m->sub_max_value[i] <= m->sub_max_value[i - 1])
        throw std::runtime_error("ContainerModel: submodels must be sorted by ascending max_value");
    if (m->sub_max_value.back() < 1.0)
      throw std::runtime_error("ContainerModel: last submodel max_value must be >= 1.0");
    for (const auto& sm : m->submodels)
    {
      if (sm->sample_rate != m->sample_rate && sm->sample_rate != -1.0 && m->sample_rate != -1.0)
      {
        std::stringstream ss;
        ss << "ContainerModel: submodel sample rate mismatch (expected " << m->sample_rate << ", got " << sm->sample_rate << ")";
        throw std::runtime_error(ss.str());
      }
      if (sm->in_channels() != 1 || sm->out_channels() != 1)
        throw std::runtime_error("SlimmableContainer: the device path needs mono submodels (the container is 1-in / 1-out)");
    }
  }
  else
    throw std::runtime_error("No config parser registered for architecture: " + arch);
  return m;
}

std::shared_ptr<ModelSpec> load_nam_text(const std::string& text, const LoadOptions& lo)
{
  Value root = json::parse(text);
  if (!root.is_object())
    throw std::runtime_error("Invalid .nam JSON: root value must be an object.");
  return build_model(root, lo);
}

double sample_rate_from_nam_text(const std::string& text)
{
  Value root = json::parse(text);
  if (root.is_object() && root.contains("sample_rate"))
    return root.at("sample_rate").as_double();
  return -1.0;
}

std::shared_ptr<ModelSpec> load_nam_file(const std::string& path, const LoadOptions& lo)
{
  // validate_nam_file nam_file.cpp:9-40
  std::ifstream in(path, std::ios::binary);
  {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f)
      throw FileValidationError("Could not validate .nam file [" + path + "]: file does not exist.");
    std::fclose(f);
  }
  if (!in.is_open())
    throw FileValidationError("Could not validate .nam file [" + path + "]: file could not be read.");
  std::stringstream ss;
  ss << in.rdbuf();
  Value root;
  try
  {
    root = json::parse(ss.str());
  }
  catch (const json::ParseError& e)
  {
    throw FileValidationError("Could not parse .nam file [" + path + "]: " + e.what());
  }
  if (!root.is_object())
    throw FileValidationError("Invalid .nam file [" + path + "]: root JSON value must be an object.");
  for (const char* key : {"version", "architecture", "config", "weights"})
    if (!root.contains(key))
      throw FileValidationError("Invalid .nam file [" + path + "]: missing required key \"" + key + "\".");
  return build_model(root, lo);
}

// ---------------------------------------------------------------------------------------------
// Slimmable (slimmable.cpp:80-294)
// ---------------------------------------------------------------------------------------------
int ratio_to_channels(double ratio, const std::vector<int>& allowed)
{
  const int idx = std::min((int)std::floor(ratio * (double)allowed.size()), (int)allowed.size() - 1);
  return allowed[std::max(idx, 0)];
}

std::vector<int> channels_for_ratio(const WaveNetSpec& full, double ratio)
{
  std::vector<int> t(full.arrays.size());
  for (size_t i = 0; i < full.arrays.size(); i++)
  {
    const auto& allowed = i < full.allowed_channels.size() ? full.allowed_channels[i] : std::vector<int>();
    t[i] = allowed.empty() ? full.arrays[i].channels : ratio_to_channels(ratio, allowed);
  }
  return t;
}

std::vector<double> slimmable_breakpoints(const WaveNetSpec& full)
{
  std::vector<double> bp;
  for (const auto& allowed : full.allowed_channels)
    for (size_t i = 1; i < allowed.size(); i++)
      bp.push_back((double)i / (double)allowed.size());
  std::sort(bp.begin(), bp.end());
  bp.erase(std::unique(bp.begin(), bp.end()), bp.end());
  return bp;
}

namespace
{
int slim_bottleneck(const LayerArraySpec& p, int new_channels)
{
  if (!p.layer1x1_active)
    return new_channels;
  return std::max(1, p.bottleneck * new_channels / p.channels);
}

struct Slicer
{
  const float* src;
  std::vector<float>& dst;
  // leading slim_out rows x slim_in cols of a row-major [full_out][full_in][k] tensor, then bias
  void conv(int full_in, int full_out, int slim_in, int slim_out, int k, bool bias)
  {
    for (int i = 0; i < full_out; i++)
      for (int j = 0; j < full_in; j++)
        for (int t = 0; t < k; t++)
        {
          const float w = *(src++);
          if (i < slim_out && j < slim_in)
            dst.push_back(w);
        }
    if (bias)
      for (int i = 0; i < full_out; i++)
      {
        const float b = *(src++);
        if (i < slim_out)
          dst.push_back(b);
      }
  }
  void copy(int n)
  {
    for (int i = 0; i < n; i++)
      dst.push_back(*(src++));
  }
};
} // namespace

WaveNetSpec slim_wavenet(const WaveNetSpec& full, const std::vector<int>& nc)
{
  if (nc.size() != full.arrays.size())
    throw std::runtime_error("SlimmableWavenet: per-array channel list size mismatch");
  WaveNetSpec s = full;
  s.slimmable = false;
  s.allowed_channels.clear();
  bool is_full = true;
  for (size_t i = 0; i < nc.size(); i++)
    is_full &= (nc[i] == full.arrays[i].channels);
  if (is_full)
    return s;

  s.weights.clear();
  Slicer sl{full.weights.data(), s.weights};
  const int na = (int)full.arrays.size();
  for (int a = 0; a < na; a++)
  {
    const LayerArraySpec& p = full.arrays[a];
    if (p.head_kernel_size != 1)
      throw std::runtime_error("SlimmableWavenet: head rechannel kernel_size must be 1 (slimming with head "
                               "kernel_size > 1 is not implemented)");
    if (p.groups_input != 1)
      throw std::runtime_error("SlimmableWavenet: groups_input > 1 not supported");
    if (p.groups_input_mixin != 1)
      throw std::runtime_error("SlimmableWavenet: groups_input_mixin > 1 not supported");
    if (p.layer1x1_active && p.layer1x1_groups != 1)
      throw std::runtime_error("SlimmableWavenet: layer1x1 groups > 1 not supported");
    if (p.head1x1_active && p.head1x1_groups != 1)
      throw std::runtime_error("SlimmableWavenet: head1x1 groups > 1 not supported");
    const int full_ch = p.channels, full_bn = p.bottleneck;
    const int slim_ch = nc[a];
    const int slim_bn = slim_bottleneck(p, slim_ch);
    const int slim_in = a == 0 ? p.input_size : nc[a - 1];
    const int slim_head = a < na - 1 ? nc[a + 1] : p.head_size;
    const int full_ho = p.head1x1_active ? p.head1x1_out : full_bn;
    const int slim_ho = p.head1x1_active ? p.head1x1_out : slim_bn;
    const int cs = p.condition_size;
    sl.conv(p.input_size, full_ch, slim_in, slim_ch, 1, false);
    for (int l = 0; l < p.num_layers(); l++)
    {
      const bool gated = p.gating_modes[l] != GATING_NONE;
      const int full_bg = gated ? 2 * full_bn : full_bn, slim_bg = gated ? 2 * slim_bn : slim_bn;
      sl.conv(full_ch, full_bg, slim_ch, slim_bg, p.kernel_sizes[l], true);
      sl.conv(cs, full_bg, cs, slim_bg, 1, false);
      if (p.layer1x1_active)
        sl.conv(full_bn, full_ch, slim_bn, slim_ch, 1, true);
      if (p.head1x1_active)
        sl.conv(full_bn, p.head1x1_out, slim_bn, p.head1x1_out, 1, true);
      auto m = [&](int k) { return p.film[k].shift ? 2 : 1; };
      if (p.film[FILM_CONV_PRE].active)
        sl.conv(cs, m(0) * full_ch, cs, m(0) * slim_ch, 1, true);
      if (p.film[FILM_CONV_POST].active)
        sl.conv(cs, m(1) * full_bg, cs, m(1) * slim_bg, 1, true);
      if (p.film[FILM_MIXIN_PRE].active)
        sl.copy(cs * m(2) * cs + m(2) * cs);
      if (p.film[FILM_MIXIN_POST].active)
        sl.conv(cs, m(3) * full_bg, cs, m(3) * slim_bg, 1, true);
      if (p.film[FILM_ACT_PRE].active)
        sl.conv(cs, m(4) * full_bg, cs, m(4) * slim_bg, 1, true);
      if (p.film[FILM_ACT_POST].active)
        sl.conv(cs, m(5) * full_bn, cs, m(5) * slim_bn, 1, true);
      if (p.film[FILM_LAYER1X1_POST].active && p.layer1x1_active)
        sl.conv(cs, m(6) * full_ch, cs, m(6) * slim_ch, 1, true);
      if (p.film[FILM_HEAD1X1_POST].active && p.head1x1_active)
        sl.copy(cs * m(7) * p.head1x1_out + m(7) * p.head1x1_out);
    }
    sl.conv(full_ho, p.head_size, slim_ho, slim_head, 1, p.head_bias);

    LayerArraySpec& q = s.arrays[a];
    q.channels = slim_ch;
    q.bottleneck = slim_bn;
    q.input_size = slim_in;
    q.head_size = slim_head;
  }
  sl.copy(1); // head_scale
  check_wavenet_weights(s);
  return s;
}

} // namespace namhip
