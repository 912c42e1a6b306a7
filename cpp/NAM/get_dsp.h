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

// Throws NamFileValidationError / std::runtime_error exactly where the reference does
// (NAM/nam_file.cpp:9-40, NAM/get_dsp.cpp:113-128, NAM/wavenet/model.cpp:671-682).
inline std::unique_ptr<DSP> get_dsp(const std::filesystem::path config_filename, DspLoadOptions options = DspLoadOptions())
{
  nam_hip_model* m = nullptr;
  detail::check(nam_hip_model_load(config_filename.string().c_str(), activations::Activation::using_fast_tanh ? 1 : 0, &m));
  return detail::wrap(m, options);
}

// The configuration-object overload (reference NAM/get_dsp.h:116 takes nlohmann::json), under the same name
inline std::unique_ptr<DSP> get_dsp(const JsonText& config, DspLoadOptions options = DspLoadOptions())
{
  nam_hip_model* m = nullptr;
  detail::check(nam_hip_model_load_json(config.text.c_str(), activations::Activation::using_fast_tanh ? 1 : 0, &m));
  return detail::wrap(m, options);
}
inline std::unique_ptr<DSP> get_dsp_json(const std::string& json_text, DspLoadOptions options = DspLoadOptions())
{
  return get_dsp(JsonText{json_text}, options);
}

} // namespace nam
