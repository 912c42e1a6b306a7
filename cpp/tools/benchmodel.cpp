// benchmodel — same protocol and command line as the reference's tools/benchmodel.cpp:23-143
// (2 s of audio in 64-frame buffers of zeros, fast tanh ON by default, prints milliseconds), running
// through the C++ adapter -> C ABI -> HIP kernels. Extra: --streams N benchmarks the batched path.
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <vector>

#include "NAM/get_dsp.h"

using std::chrono::duration;
using std::chrono::duration_cast;
using std::chrono::high_resolution_clock;
using std::chrono::milliseconds;

#define AUDIO_BUFFER_SIZE 64

int main(int argc, char* argv[])
{
  if (argc < 2)
  {
    std::cerr << "Usage: benchmodel <model_path> [--slim <0..1>] [--no-fast-tanh] [--streams N]\n";
    return 1;
  }
  const char* modelPath = argv[1];
  double slim = -1.0;
  bool fast_tanh = true;
  int streams = 1;
  for (int i = 2; i < argc; i++)
  {
    if (!std::strcmp(argv[i], "--slim") && i + 1 < argc)
      slim = std::atof(argv[++i]);
    else if (!std::strcmp(argv[i], "--no-fast-tanh"))
      fast_tanh = false;
    else if (!std::strcmp(argv[i], "--streams") && i + 1 < argc)
      streams = std::atoi(argv[++i]);
  }
  if (fast_tanh)
    nam::activations::Activation::enable_fast_tanh();
  try
  {
    std::cout << "Loading model " << modelPath << "\n";
    const size_t numBuffers = (48000 / AUDIO_BUFFER_SIZE) * 2;
    if (streams <= 1)
    {
      auto model = nam::get_dsp(modelPath);
      if (slim >= 0.0)
        if (auto* s = dynamic_cast<nam::SlimmableModel*>(model.get()))
          s->SetSlimmableSize(slim);
      model->Reset(model->GetExpectedSampleRate(), AUDIO_BUFFER_SIZE);
      const int ic = model->NumInputChannels(), oc = model->NumOutputChannels();
      std::vector<std::vector<NAM_SAMPLE>> in(ic, std::vector<NAM_SAMPLE>(AUDIO_BUFFER_SIZE, 0.0)),
        out(oc, std::vector<NAM_SAMPLE>(AUDIO_BUFFER_SIZE, 0.0));
      std::vector<NAM_SAMPLE*> inp(ic), outp(oc);
      for (int c = 0; c < ic; c++)
        inp[c] = in[c].data();
      for (int c = 0; c < oc; c++)
        outp[c] = out[c].data();
      std::cout << "Running benchmark\n";
      auto t1 = high_resolution_clock::now();
      for (size_t i = 0; i < numBuffers; i++)
        model->process(inp.data(), outp.data(), AUDIO_BUFFER_SIZE);
      auto t2 = high_resolution_clock::now();
      duration<double, std::milli> ms = t2 - t1;
      std::cout << duration_cast<milliseconds>(t2 - t1).count() << "ms\n" << ms.count() << "ms\n";
    }
    else
    {
      nam_hip_model* raw = nullptr;
      nam::detail::check(nam_hip_model_load(modelPath, fast_tanh ? 1 : 0, &raw));
      std::shared_ptr<nam_hip_model> m(raw, nam::detail::ModelDeleter());
      nam::BatchDSP batch(m, streams);
      batch.Reset(48000.0, AUDIO_BUFFER_SIZE);
      std::vector<float> in((size_t)streams * batch.NumInputChannels() * AUDIO_BUFFER_SIZE, 0.0f),
        out((size_t)streams * batch.NumOutputChannels() * AUDIO_BUFFER_SIZE, 0.0f);
      std::cout << "Running benchmark (" << streams << " streams, host buffers)\n";
      auto t1 = high_resolution_clock::now();
      for (size_t i = 0; i < numBuffers; i++)
        batch.process_batch(in.data(), out.data(), AUDIO_BUFFER_SIZE);
      auto t2 = high_resolution_clock::now();
      duration<double, std::milli> ms = t2 - t1;
      std::cout << ms.count() << "ms for 2 s x " << streams << " streams = " << 2000.0 * streams / ms.count()
                << " x real time\n";
    }
  }
  catch (const std::exception& e)
  {
    std::cerr << "Error: " << e.what() << "\n";
    return 1;
  }
  return 0;
}
