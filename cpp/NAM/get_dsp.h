// NAM/get_dsp.h — `nam::get_dsp(path)` on top of the nam_hip C ABI (reference NAM/get_dsp.h:85-116).
#pragma once

#include <filesystem>
#include <optional>

#include "dsp.h"

namespace nam
{

const std::string LATEST_FULLY_SUPPORTED_NAM_FILE_VERSION = "0.7.0";
const std::string EARLIEST_SUPPORTED_NAM_FILE_VERSION = "0.5.0";

struct DspLoadOptions // reference NAM/get_dsp.h:70-78
{
  std::optional<bool> prewarm = std::nullopt;
};

namespace detail
{
inline std::unique_ptr<DSP> wrap(nam_hip_model* raw, const DspLoadOptions& options)
{
  std::shared_ptr<nam_hip_model> model(raw, ModelDeleter());
  nam_hip_model_info info{};
  check(nam_hip_model_get_info(model.get(), &info));
  std::unique_ptr<DSP> dsp;
  if (info.is_slimmable)
    dsp = std::make_unique<SlimmableDSP>(model);
  else
    dsp = std::make_unique<DSP>(model);
  (void)options; // load-time prewarm override only affects objects constructed during loading
  return dsp;
}
} // namespace detail

// Throws NamFileValidationError / std::runtime_error exactly where the reference does
// (NAM/nam_file.cpp:9-40, NAM/get_dsp.cpp:113-128, NAM/wavenet/model.cpp:671-682).
inline std::unique_ptr<DSP> get_dsp(const std::filesystem::path config_filename, DspLoadOptions options = DspLoadOptions())
{
  nam_hip_model* m = nullptr;
  detail::check(nam_hip_model_load(config_filename.string().c_str(), activations::Activation::using_fast_tanh ? 1 : 0, &m));
  return detail::wrap(m, options);
}

// JSON-text overload (the reference takes nlohmann::json; callers holding a json object pass j.dump()).
inline std::unique_ptr<DSP> get_dsp_json(const std::string& json_text, DspLoadOptions options = DspLoadOptions())
{
  nam_hip_model* m = nullptr;
  detail::check(nam_hip_model_load_json(json_text.c_str(), activations::Activation::using_fast_tanh ? 1 : 0, &m));
  return detail::wrap(m, options);
}

} // namespace nam
