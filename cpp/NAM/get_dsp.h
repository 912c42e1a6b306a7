// NAM/get_dsp.h — `nam::get_dsp(path)` on top of the nam_hip C ABI (reference NAM/get_dsp.h:85-116).
#pragma once

#include <filesystem>
#include <iostream>
#include <mutex>
#include <optional>
#include <sstream>

#include "dsp.h"

namespace nam
{

const std::string LATEST_FULLY_SUPPORTED_NAM_FILE_VERSION = "0.7.0";
const std::string EARLIEST_SUPPORTED_NAM_FILE_VERSION = "0.5.0";

// ---- the version gate with caller-registered checkers (reference NAM/get_dsp.h:12-64, get_dsp.cpp:18-128) ----
enum class Supported
{
  NO = 0,
  PARTIAL = 1,
  YES = 2
};
class IVersionSupportChecker
{
public:
  virtual ~IVersionSupportChecker() = default;
  virtual Supported support(const std::string& version) const = 0;
};
namespace detail
{
inline std::vector<std::shared_ptr<const IVersionSupportChecker>>& version_support_registry()
{
  static std::vector<std::shared_ptr<const IVersionSupportChecker>> registry; // (the core checker lives in the library)
  return registry;
}
inline std::mutex& version_support_registry_mutex()
{
  static std::mutex m;
  return m;
}
} // namespace detail
inline void register_version_support_checker(std::shared_ptr<const IVersionSupportChecker> checker)
{
  if (!checker)
    throw std::invalid_argument("version support checker cannot be null");
  std::lock_guard<std::mutex> lock(detail::version_support_registry_mutex());
  detail::version_support_registry().push_back(std::move(checker));
}
// the best answer of the built-in checker (nam_hip_version_support) and every registered one
inline Supported is_version_supported(const std::string version)
{
  int best = nam_hip_version_support(version.c_str());
  std::lock_guard<std::mutex> lock(detail::version_support_registry_mutex());
  for (const auto& checker : detail::version_support_registry())
    best = std::max(best, static_cast<int>(checker->support(version)));
  return static_cast<Supported>(best);
}
inline void verify_config_version(const std::string versionStr)
{
  const Supported support = is_version_supported(versionStr);
  if (support == Supported::NO)
    throw std::runtime_error("Model config is an unsupported version " + versionStr + ".");
  if (support == Supported::PARTIAL)
    std::cerr << "Model config is a partially-supported version " << versionStr << ". Continuing with partial support."
              << std::endl;
}

struct DspLoadOptions // reference NAM/get_dsp.h:70-78
{
  std::optional<bool> prewarm = std::nullopt;
};

// The text of a .nam file handed over in memory. The reference's overloads take nlohmann::json (NAM/get_dsp.h:110-116);
// this library parses the text itself, so a caller holding a json object passes JsonText{j.dump()}. (A distinct type
// rather than std::string_view: get_dsp("model.nam") must keep meaning "the file at that path".)
struct JsonText
{
  std::string text;
};

namespace detail
{
inline std::unique_ptr<DSP> wrap_with_current_default(nam_hip_model* raw)
{
  std::shared_ptr<nam_hip_model> model(raw, ModelDeleter());
  nam_hip_model_info info{};
  check(nam_hip_model_get_info(model.get(), &info));
  if (info.is_slimmable)
    return std::make_unique<SlimmableDSP>(model);
  return std::make_unique<DSP>(model);
}
// DspLoadOptions::prewarm as the reference applies it (NAM/get_dsp.cpp:263-273): the override is the thread-local
// default WHILE the object is constructed, and the returned object goes back to the caller's previous default
inline std::unique_ptr<DSP> wrap(nam_hip_model* raw, const DspLoadOptions& options)
{
  if (!options.prewarm.has_value())
    return wrap_with_current_default(raw);
  ScopedPrewarmOnResetDefault scoped(*options.prewarm);
  auto dsp = wrap_with_current_default(raw);
  if (dsp != nullptr)
    dsp->SetPrewarmOnReset(scoped.PreviousPrewarmOnReset());
  return dsp;
}
} // namespace detail

// nam::dspData (reference NAM/dsp.h:348-357). The reference holds `config` and `metadata` as nlohmann::json; this
// library does not depend on nlohmann, so they are the same values as JSON TEXT (a caller holding json objects uses
// j.dump() / nlohmann::json::parse(text)). An empty `metadata` means none ("null").
struct dspData
{
  std::string version;
  std::string architecture;
  std::string config; // JSON text of the "config" value
  std::string metadata; // JSON text of the "metadata" value
  std::vector<float> weights;
  double expected_sample_rate = NAM_UNKNOWN_EXPECTED_SAMPLE_RATE;
};

namespace detail
{
inline std::string model_string(const nam_hip_model* m, int field)
{
  const int64_t n = nam_hip_model_get_string(m, field, nullptr, 0);
  if (n < 0)
    check((int)n);
  std::string s((size_t)n, '\0');
  nam_hip_model_get_string(m, field, s.data(), n + 1);
  return s;
}
// populate_dsp_data (reference NAM/get_dsp.cpp:141-154) from the loaded model
inline void populate_dsp_data(const nam_hip_model* m, dspData& out)
{
  nam_hip_model_info info{};
  check(nam_hip_model_get_info(m, &info));
  out.version = model_string(m, NAM_HIP_FIELD_VERSION);
  out.architecture = model_string(m, NAM_HIP_FIELD_ARCHITECTURE);
  out.config = model_string(m, NAM_HIP_FIELD_CONFIG_JSON);
  out.metadata = model_string(m, NAM_HIP_FIELD_METADATA_JSON);
  const int64_t n = nam_hip_model_get_weights(m, nullptr, 0);
  out.weights.assign((size_t)std::max<int64_t>(n, 0), 0.0f);
  if (n > 0)
    nam_hip_model_get_weights(m, out.weights.data(), n);
  out.expected_sample_rate = info.expected_sample_rate;
}
// The top-level "version" of a .nam document, for the caller-side gate (only consulted when checkers are registered).
inline bool has_custom_checkers()
{
  std::lock_guard<std::mutex> lock(version_support_registry_mutex());
  return !version_support_registry().empty();
}
inline nam_hip_load_options load_options(bool version_checked)
{
  nam_hip_load_options o = NAM_HIP_LOAD_OPTIONS_INIT;
  o.fast_tanh = activations::Activation::using_fast_tanh ? 1 : 0;
  o.version_checked_by_caller = version_checked ? 1 : 0;
  return o;
}
} // namespace detail

inline std::unique_ptr<DSP> get_dsp(const std::filesystem::path config_filename, dspData& returnedConfig, DspLoadOptions options);
inline std::unique_ptr<DSP> get_dsp(const JsonText& config, dspData& returnedConfig, DspLoadOptions options);

// Throws NamFileValidationError / std::runtime_error exactly where the reference does
// (NAM/nam_file.cpp:9-40, NAM/get_dsp.cpp:113-128, NAM/wavenet/model.cpp:671-682).
inline std::unique_ptr<DSP> get_dsp(const std::filesystem::path config_filename, DspLoadOptions options = DspLoadOptions())
{
  if (detail::has_custom_checkers()) // the caller-side version gate decides: the dspData route runs it (get_dsp.cpp:156-160)
  {
    dspData temp;
    return get_dsp(config_filename, temp, options);
  }
  nam_hip_model* m = nullptr;
  detail::check(nam_hip_model_load(config_filename.string().c_str(), activations::Activation::using_fast_tanh ? 1 : 0, &m));
  return detail::wrap(m, options);
}

// The configuration-object overload (reference NAM/get_dsp.h:116 takes nlohmann::json), under the same name
inline std::unique_ptr<DSP> get_dsp(const JsonText& config, DspLoadOptions options = DspLoadOptions())
{
  if (detail::has_custom_checkers())
  {
    dspData temp;
    return get_dsp(config, temp, options);
  }
  nam_hip_model* m = nullptr;
  detail::check(nam_hip_model_load_json(config.text.c_str(), activations::Activation::using_fast_tanh ? 1 : 0, &m));
  return detail::wrap(m, options);
}
inline std::unique_ptr<DSP> get_dsp_json(const std::string& json_text, DspLoadOptions options = DspLoadOptions())
{
  return get_dsp(JsonText{json_text}, options);
}

} // namespace nam

namespace nam
{
// get_dsp(dspData& conf) — reference NAM/get_dsp.h:91, get_dsp.cpp:232-273: verify_config_version (registered checkers
// included), the architecture's config parser, weight binding, metadata.
inline std::unique_ptr<DSP> get_dsp(dspData& conf, DspLoadOptions options = DspLoadOptions())
{
  verify_config_version(conf.version);
  const nam_hip_load_options lo = detail::load_options(true);
  nam_hip_model* m = nullptr;
  detail::check(nam_hip_model_load_parts(conf.version.c_str(), conf.architecture.c_str(), conf.config.c_str(),
                                         conf.metadata.empty() ? nullptr : conf.metadata.c_str(), conf.weights.data(),
                                         (int64_t)conf.weights.size(), conf.expected_sample_rate, &lo, &m));
  return detail::wrap(m, options);
}

// get_dsp(path, dspData& returnedConfig) — reference NAM/get_dsp.h:101, get_dsp.cpp:168-181
inline std::unique_ptr<DSP> get_dsp(const std::filesystem::path config_filename, dspData& returnedConfig,
                                    DspLoadOptions options = DspLoadOptions())
{
  // with registered checkers the caller-side gate decides (it can accept versions the built-in one refuses): load
  // ungated, then verify the version the document carries, as populate_dsp_data does first (get_dsp.cpp:143)
  const bool custom = detail::has_custom_checkers();
  const nam_hip_load_options lo = detail::load_options(custom);
  nam_hip_model* m = nullptr;
  detail::check(nam_hip_model_load_ex(config_filename.string().c_str(), nullptr, &lo, &m));
  std::shared_ptr<nam_hip_model> keep(m, detail::ModelDeleter());
  if (custom)
    verify_config_version(detail::model_string(m, NAM_HIP_FIELD_VERSION));
  detail::populate_dsp_data(m, returnedConfig);
  dspData conf = returnedConfig; // (the reference builds the object from a copy, get_dsp.cpp:172-180)
  return get_dsp(conf, options);
}

// get_dsp(json, dspData& returnedConfig) — reference NAM/get_dsp.h:109, get_dsp.cpp:183-196
inline std::unique_ptr<DSP> get_dsp(const JsonText& config, dspData& returnedConfig, DspLoadOptions options = DspLoadOptions())
{
  const bool custom = detail::has_custom_checkers();
  const nam_hip_load_options lo = detail::load_options(custom);
  nam_hip_model* m = nullptr;
  detail::check(nam_hip_model_load_ex(nullptr, config.text.c_str(), &lo, &m));
  std::shared_ptr<nam_hip_model> keep(m, detail::ModelDeleter());
  if (custom)
    verify_config_version(detail::model_string(m, NAM_HIP_FIELD_VERSION));
  detail::populate_dsp_data(m, returnedConfig);
  dspData conf = returnedConfig;
  return get_dsp(conf, options);
}

// get_sample_rate_from_nam_file — reference NAM/get_dsp.h:121, get_dsp.cpp:275-281 (takes the parsed document there)
inline double get_sample_rate_from_nam_file(const JsonText& j)
{
  double sr = NAM_UNKNOWN_EXPECTED_SAMPLE_RATE;
  detail::check(nam_hip_sample_rate_from_nam(nullptr, j.text.c_str(), &sr));
  return sr;
}
inline double get_sample_rate_from_nam_file(const std::filesystem::path& filename)
{
  double sr = NAM_UNKNOWN_EXPECTED_SAMPLE_RATE;
  detail::check(nam_hip_sample_rate_from_nam(filename.string().c_str(), nullptr, &sr));
  return sr;
}

#ifdef NLOHMANN_JSON_VERSION_MAJOR
// For callers that include nlohmann/json.hpp before this header (the plugin does): the reference's own signatures.
inline std::unique_ptr<DSP> get_dsp(const nlohmann::json& config, DspLoadOptions options = DspLoadOptions())
{
  return get_dsp(JsonText{config.dump()}, options);
}
inline std::unique_ptr<DSP> get_dsp(const nlohmann::json& config, dspData& returnedConfig, DspLoadOptions options = DspLoadOptions())
{
  return get_dsp(JsonText{config.dump()}, returnedConfig, options);
}
inline double get_sample_rate_from_nam_file(const nlohmann::json& j)
{
  return get_sample_rate_from_nam_file(JsonText{j.dump()});
}
#endif
} // namespace nam
