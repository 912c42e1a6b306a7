// model_spec.h — typed, architecture-tagged description of a loaded .nam model (host side).
//
// This is the "model IR" that sits between the .nam JSON (reference schema:
// NAM/wavenet/model.cpp:913-1276, NAM/lstm.cpp:170-181, NAM/get_dsp.cpp:141-154) and the
// device plan (plan.h). Nothing here touches the GPU.
#pragma once

#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace namhip
{

// Same exception split as the reference so the C++ adapter can re-throw identically:
// nam::NamFileValidationError (NAM/nam_file.h:11) vs std::runtime_error for everything else.
struct FileValidationError : std::runtime_error
{
  explicit FileValidationError(const std::string& m)
  : std::runtime_error(m)
  {
  }
};

enum ActType : int
{
  ACT_IDENTITY = 0,
  ACT_TANH = 1,
  ACT_HARDTANH = 2,
  ACT_FASTTANH = 3,
  ACT_RELU = 4,
  ACT_LEAKYRELU = 5,
  ACT_PRELU = 6,
  ACT_SIGMOID = 7,
  ACT_SILU = 8,
  ACT_HARDSWISH = 9,
  ACT_LEAKYHARDTANH = 10,
  ACT_SOFTSIGN = 11,
  // LSTM-only helper (activations::fast_sigmoid, NAM/activations.h:100-103)
  ACT_FASTSIGMOID = 12,
  // FastLUTActivation (NAM/activations.h:371-422): p = {min_x, max_x, inv_step, n_points}, `slopes` = the table
  ACT_LUT = 13
};

struct ActSpec
{
  int type = ACT_IDENTITY;
  float p[4] = {0, 0, 0, 0}; // LeakyReLU: p[0]; LeakyHardtanh: min_val,max_val,min_slope,max_slope
  std::vector<float> slopes; // PReLU slopes; ACT_LUT: the lookup table
};

// Activation::enable_lut(function_name, min, max, n_points) — NAM/activations.cpp:189-212 — as a load option
struct LutSpec
{
  int act_type = 0; // ACT_TANH, ACT_SIGMOID or ACT_SILU (the only three the reference accepts)
  float min_x = 0.0f, max_x = 0.0f;
  int n_points = 0;
};
// What the reference keeps in process-globals around get_dsp (activations.cpp:168-212), as explicit load options
struct LoadOptions
{
  bool fast_tanh = false;
  std::vector<LutSpec> luts;
  bool skip_version_gate = false; // the caller ran verify_config_version with its own checkers (get_dsp.h:60)
};

enum GatingMode : int
{
  GATING_NONE = 0,
  GATING_GATED = 1,
  GATING_BLENDED = 2
};

struct FilmSpec
{
  bool active = false;
  bool shift = false;
  int groups = 1;
};

enum FilmSlot : int
{
  FILM_CONV_PRE = 0,
  FILM_CONV_POST,
  FILM_MIXIN_PRE,
  FILM_MIXIN_POST,
  FILM_ACT_PRE,
  FILM_ACT_POST,
  FILM_LAYER1X1_POST,
  FILM_HEAD1X1_POST,
  FILM_COUNT
};

// One entry of config["layers"] — nam::wavenet::LayerArrayParams (NAM/wavenet/params.h)
struct LayerArraySpec
{
  int input_size = 0;
  int condition_size = 0;
  int head_size = 0;
  int head_kernel_size = 1;
  int head_dilation = 1;
  bool head_bias = false;
  int channels = 0;
  int bottleneck = 0;
  std::vector<int> kernel_sizes;
  std::vector<int> dilations;
  std::vector<ActSpec> activations;
  std::vector<int> gating_modes;
  std::vector<ActSpec> secondary_activations;
  int groups_input = 1;
  int groups_input_mixin = 1;
  bool layer1x1_active = true;
  int layer1x1_groups = 1;
  bool head1x1_active = false;
  int head1x1_out = 0;
  int head1x1_groups = 1;
  FilmSpec film[FILM_COUNT];

  int num_layers() const { return (int)dilations.size(); }
  int head_output_size() const { return head1x1_active ? head1x1_out : bottleneck; } // model.cpp:399-401
};

struct PostHeadSpec // nam::wavenet::HeadParams, model.cpp:21-44
{
  int in_channels = 0;
  int channels = 0;
  int out_channels = 0;
  std::vector<int> kernel_sizes;
  ActSpec activation;
};

struct ModelSpec;

struct WaveNetSpec
{
  int in_channels = 1;
  std::vector<LayerArraySpec> arrays;
  bool with_head = false;
  PostHeadSpec head;
  float head_scale_json = 0.0f; // the JSON field; the effective value is the last weight (model.cpp:670)
  std::shared_ptr<ModelSpec> condition_dsp; // nested model (must be a WaveNet for the device path)
  std::vector<float> weights;

  // Slimmable (NAM/wavenet/slimmable.cpp): per-array allowed channel counts; empty = not slimmable
  bool slimmable = false;
  std::vector<std::vector<int>> allowed_channels;

  int out_channels() const { return with_head ? head.out_channels : arrays.back().head_size; }
  long expected_weight_count() const;
  int prewarm_samples() const; // model.cpp:653-658
};

struct LSTMSpec
{
  int num_layers = 0;
  int input_size = 0;
  int hidden_size = 0;
  int in_channels = 1;
  int out_channels = 1;
  std::vector<float> weights;
  long expected_weight_count() const;
};

enum Arch : int
{
  ARCH_WAVENET = 1,
  ARCH_LSTM = 2,
  ARCH_CONTAINER = 3 // SlimmableContainer (NAM/container.cpp): submodels of different sizes, one active at a time
};

struct ModelSpec
{
  int arch = 0;
  std::string version;
  double sample_rate = -1.0; // NAM_UNKNOWN_EXPECTED_SAMPLE_RATE (dsp.h:28)
  bool has_loudness = false, has_input_level = false, has_output_level = false;
  double loudness = 0.0, input_level = 0.0, output_level = 0.0;
  bool fast_tanh = false; // resolved at load time (the reference uses a process-global, activations.cpp:168)
  // what nam::dspData carries besides the weights (NAM/dsp.h:348-357, filled by populate_dsp_data get_dsp.cpp:141-154):
  // the architecture string and the "config" / "metadata" values as JSON text ("null" when the file has no metadata)
  std::string architecture_name, config_text, metadata_text = "null";
  WaveNetSpec wavenet;
  LSTMSpec lstm;
  // ARCH_CONTAINER: submodels sorted by ascending max_value; SetSlimmableSize(v) activates the first one with
  // v < max_value, else the last (container.cpp:103-115). The container itself is always 1-in / 1-out
  // (container.cpp:19) and has no weights of its own.
  std::vector<double> sub_max_value;
  std::vector<std::shared_ptr<ModelSpec>> submodels;

  int in_channels() const { return arch == ARCH_WAVENET ? wavenet.in_channels : arch == ARCH_LSTM ? lstm.in_channels : 1; }
  int out_channels() const { return arch == ARCH_WAVENET ? wavenet.out_channels() : arch == ARCH_LSTM ? lstm.out_channels : 1; }
  int prewarm_samples() const;
  int container_index(double val) const; // container.cpp:103-115
};

// ---- loader entry points (nam_loader.cpp) ----
// nam::get_dsp(path) front half: validate_nam_file + populate_dsp_data + config parse.
std::shared_ptr<ModelSpec> load_nam_file(const std::string& path, const LoadOptions& lo);
std::shared_ptr<ModelSpec> load_nam_text(const std::string& json_text, const LoadOptions& lo);
// get_sample_rate_from_nam_file (get_dsp.cpp:275-281) on the text of a .nam document: "sample_rate" or -1.0
double sample_rate_from_nam_text(const std::string& json_text);

// Slimmable helpers (slimmable.cpp:80-294)
int ratio_to_channels(double ratio, const std::vector<int>& allowed);
std::vector<int> channels_for_ratio(const WaveNetSpec& full, double ratio);
// Returns a plain (non-slimmable) WaveNetSpec of the requested per-array widths.
WaveNetSpec slim_wavenet(const WaveNetSpec& full, const std::vector<int>& new_channels);
std::vector<double> slimmable_breakpoints(const WaveNetSpec& full);

// Version gate (get_dsp.cpp:18-39,113-128): 0 = no, 1 = partial, 2 = yes
int version_support(const std::string& version);

} // namespace namhip
