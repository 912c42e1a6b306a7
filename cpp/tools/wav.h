// wav.h — the slice of AudioDSPTools' dsp::wav interface the reference's tools/render.cpp names (render.cpp:15, 129-136:
// Load, LoadReturnCode, GetMsgForLoadReturnCode), over this repository's own reader (wav_io.h). AudioDSPTools is an
// un-vendored submodule of the reference (Dependencies/AudioDSPTools, empty here), so the signatures are the ones its call
// site fixes: with -Icpp -Icpp/tools the reference's render.cpp compiles unmodified against the adapter.
#pragma once

#include <string>
#include <vector>

#include "wav_io.h"

namespace dsp
{
namespace wav
{
enum class LoadReturnCode
{
  SUCCESS = 0,
  ERROR_OPENING,
  ERROR_NOT_WAV,
  ERROR_UNSUPPORTED_FORMAT,
  ERROR_OTHER
};

namespace detail
{
inline std::string& last_message()
{
  static thread_local std::string msg;
  return msg;
}
} // namespace detail

// mono PCM 16 / 24 / 32-bit or IEEE float 32-bit -> float samples in [-1, 1) and the file's sample rate
inline LoadReturnCode Load(const char* fileName, std::vector<float>& audio, double& sampleRate)
{
  try
  {
    wavio::Audio a = wavio::load(fileName);
    audio = std::move(a.samples);
    sampleRate = a.sample_rate;
    detail::last_message().clear();
    return LoadReturnCode::SUCCESS;
  }
  catch (const std::exception& e)
  {
    const std::string what = e.what();
    detail::last_message() = what;
    if (what.find("cannot open") != std::string::npos)
      return LoadReturnCode::ERROR_OPENING;
    if (what.find("not a RIFF") != std::string::npos)
      return LoadReturnCode::ERROR_NOT_WAV;
    if (what.find("unsupported") != std::string::npos || what.find("only mono") != std::string::npos)
      return LoadReturnCode::ERROR_UNSUPPORTED_FORMAT;
    return LoadReturnCode::ERROR_OTHER;
  }
}

inline std::string GetMsgForLoadReturnCode(LoadReturnCode rc)
{
  if (rc == LoadReturnCode::SUCCESS)
    return "success";
  const std::string& m = detail::last_message();
  if (!m.empty())
    return m;
  switch (rc)
  {
    case LoadReturnCode::ERROR_OPENING: return "could not open the file";
    case LoadReturnCode::ERROR_NOT_WAV: return "not a RIFF/WAVE file";
    case LoadReturnCode::ERROR_UNSUPPORTED_FORMAT: return "unsupported sample format";
    default: return "could not read the file";
  }
}
} // namespace wav
} // namespace dsp
