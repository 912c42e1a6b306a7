// render — the reference's offline re-amp tool (tools/render.cpp:63-205) on the MI355X batch path.
//   render [--slim <0..1>] <model.nam> <input.wav> [output.wav]            (the reference's command line)
//   render [--slim <0..1>] [--devices <list>] [--plan-only] <model.nam> --batch <out_dir> <in1.wav> <in2.wav> ...
// --devices ("all", "0-7", "0,2,5"; duplicates allowed): the files are dealt to the listed GPUs, longest first in snake
// order, and every GPU renders its share as its own batch on its own host thread (cpp/NAM/multi_device.h; no collective:
// streams never interact). --plan-only prints that assignment ("device <d>: <file> (<frames> frames)") and exits
// without touching a GPU (--device-count N stands in for the visible devices).
// The batch form pushes N files through one model as N independent streams of one GPU batch (files may have
// different lengths); outputs are <out_dir>/<input stem>.wav. As in the reference: mono input only, the WAV's
// sample rate must match the model's expected rate when it has one, Reset(rate, 64) with the model's prewarm,
// output channel 0 written as mono float32. The 64-frame block loop of the reference becomes one resident
// launch over device-resident audio (nam_hip_batch_render_f32); results do not depend on the partition.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <iostream>
#include <string>
#include <vector>

#include "NAM/get_dsp.h"
#include "NAM/multi_device.h"
#include "wav_io.h"

int main(int argc, char* argv[])
{
  // options first-come-first-served, everything else is positional
  struct Options
  {
    bool slim = false;
    double slimValue = -1.0;
    std::string batchDir;
    std::string devices; // empty: device 0, one batch
    bool planOnly = false;
    int deviceCount = -1; // --device-count: stands in for nam_hip_device_count (plan-only runs on boxes without a GPU)
    std::vector<std::string> positional;
  } opt;
  auto value_of = [&](int& i, const char* what) -> const char* {
    if (i + 1 >= argc)
    {
      std::cerr << "Error: " << what << "\n";
      std::exit(1);
    }
    return argv[++i];
  };
  for (int i = 1; i < argc; i++)
  {
    if (!std::strcmp(argv[i], "--slim"))
    {
      const char* text = value_of(i, "--slim requires a value between 0.0 and 1.0");
      char* rest = nullptr;
      opt.slimValue = std::strtod(text, &rest);
      if (rest == text || *rest || !(opt.slimValue >= 0.0 && opt.slimValue <= 1.0))
      {
        std::cerr << "Error: --slim value must be a number between 0.0 and 1.0\n";
        return 1;
      }
      opt.slim = true;
    }
    else if (!std::strcmp(argv[i], "--batch"))
      opt.batchDir = value_of(i, "--batch requires an output directory");
    else if (!std::strcmp(argv[i], "--devices"))
      opt.devices = value_of(i, "--devices requires a list such as all, 0-7 or 0,2,5");
    else if (!std::strcmp(argv[i], "--device-count"))
      opt.deviceCount = std::atoi(value_of(i, "--device-count requires a number"));
    else if (!std::strcmp(argv[i], "--plan-only"))
      opt.planOnly = true;
    else
      opt.positional.emplace_back(argv[i]);
  }
  const bool hasSlim = opt.slim;
  const double slimValue = opt.slimValue;
  const std::string& batchDir = opt.batchDir;
  const std::vector<std::string>& pos = opt.positional;
  const bool batchMode = !batchDir.empty();
  if ((!batchMode && (pos.size() < 2 || pos.size() > 3)) || (batchMode && pos.size() < 2))
  {
    std::cerr << "Usage: render [--slim <0.0-1.0>] <model.nam> <input.wav> [output.wav]\n"
                 "       render [--slim <0.0-1.0>] [--devices <all|0-7|0,2,5>] [--plan-only] <model.nam> --batch <out_dir> <in1.wav> "
                 "[<in2.wav> ...]\n";
    return 1;
  }
  try
  {
    const std::string modelPath = pos[0];
    std::vector<std::string> inputs, outputs;
    if (batchMode)
    {
      std::filesystem::create_directories(batchDir);
      for (size_t i = 1; i < pos.size(); i++)
      {
        inputs.push_back(pos[i]);
        outputs.push_back((std::filesystem::path(batchDir) / (std::filesystem::path(pos[i]).stem().string() + ".wav")).string());
      }
    }
    else
    {
      inputs.push_back(pos[1]);
      outputs.push_back(pos.size() >= 3 ? pos[2] : "output.wav");
    }
    const int n = (int)inputs.size();

    // the devices the batch is spread over (multi_device.h); default: one batch on device 0
    std::vector<int> devices{0};
    if (!opt.devices.empty())
    {
      int count = opt.deviceCount;
      if (count < 0)
        nam::detail::check(nam_hip_device_count(&count));
      devices = nam::parse_device_list(opt.devices, count);
    }
    if (opt.planOnly)
    {
      std::vector<int64_t> lengths;
      for (const auto& f : inputs)
        lengths.push_back((int64_t)wavio::load(f).samples.size());
      const auto deal = nam::deal_by_length(lengths, (int)devices.size());
      for (size_t d = 0; d < devices.size(); d++)
        for (int i : deal[d])
          std::cout << "device " << devices[d] << " (batch " << d << "): " << inputs[(size_t)i] << " (" << lengths[(size_t)i] << " frames)\n";
      return 0;
    }

    std::cerr << "Loading model [" << modelPath << "]\n";
    nam_hip_model* raw = nullptr;
    nam::detail::check(nam_hip_model_load(modelPath.c_str(), nam::activations::Activation::using_fast_tanh ? 1 : 0, &raw));
    std::shared_ptr<nam_hip_model> model(raw, nam::detail::ModelDeleter());
    nam::BatchDSP dsp(model, n);
    std::cerr << "Model loaded successfully\n";
    if (dsp.NumInputChannels() != 1)
    {
      std::cerr << "Error: render tool currently supports mono input only (model has " << dsp.NumInputChannels()
                << " input channels)\n";
      return 1;
    }

    std::vector<wavio::Audio> audio(n);
    const double expectedRate = dsp.GetExpectedSampleRate();
    double sampleRate = expectedRate;
    for (int i = 0; i < n; i++)
    {
      audio[i] = wavio::load(inputs[i]);
      if (expectedRate > 0 && std::abs(audio[i].sample_rate - expectedRate) > 0.5)
      {
        std::cerr << "Error: Input WAV sample rate (" << audio[i].sample_rate << " Hz) does not match model expected rate ("
                  << expectedRate << " Hz)\n";
        return 1;
      }
      if (expectedRate <= 0)
      {
        if (i > 0 && std::abs(audio[i].sample_rate - sampleRate) > 0.5)
        {
          std::cerr << "Error: the input files of one batch must share a sample rate\n";
          return 1;
        }
        sampleRate = audio[i].sample_rate;
      }
    }

    const bool spread = devices.size() > 1 || devices[0] != 0; // several batches, one per listed device (multi_device.h)
    if (hasSlim && !dsp.IsSlimmable())
    {
      std::cerr << "Error: --slim requires a model that implements the SlimmableModel interface\n";
      return 1;
    }
    if (!spread)
    {
      dsp.Reset(sampleRate, 64); // bufferSize 64 as the reference (tools/render.cpp:146-147): fixes the prewarm length
      if (hasSlim)
      {
        std::cerr << "Setting slimmable size to " << slimValue << "\n";
        dsp.SetSlimmableSize(nullptr, 0, slimValue);
      }
    }

    const int oc = dsp.NumOutputChannels();
    std::vector<std::vector<float>> out(n);
    std::vector<const float*> inp(n);
    std::vector<float*> outp(n);
    std::vector<int64_t> frames(n);
    for (int i = 0; i < n; i++)
    {
      frames[i] = (int64_t)audio[i].samples.size();
      out[i].resize((size_t)oc * audio[i].samples.size());
      inp[i] = audio[i].samples.data();
      outp[i] = out[i].data();
    }
    if (spread)
      nam::render_on_devices(model, devices, inp.data(), outp.data(), frames.data(), n, sampleRate, hasSlim ? slimValue : -1.0);
    else
      dsp.render(inp.data(), outp.data(), frames.data());
    for (int i = 0; i < n; i++)
    {
      wavio::save_float32(outputs[i], out[i].data(), audio[i].samples.size(), sampleRate); // channel 0
      std::cerr << "Wrote " << audio[i].samples.size() << " samples to " << outputs[i] << "\n";
    }
  }
  catch (const std::exception& e)
  {
    std::cerr << "Error: " << e.what() << "\n";
    return 1;
  }
  return 0;
}
