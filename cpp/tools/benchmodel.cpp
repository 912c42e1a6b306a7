// benchmodel — same protocol and command line as the reference's tools/benchmodel.cpp:23-143
// (2 s of audio in 64-frame buffers of zeros, fast tanh ON by default, prints milliseconds), running
// through the C++ adapter -> C ABI -> HIP kernels. Extras: --streams N benchmarks the batched path; the per-buffer round
// trip (host buffer in -> host buffer out) is printed as min / p50 / p99 / max; --count-allocs counts heap allocations
// made during the timed loop, by the module that asked for them (the reference's real-time-safety check,
// tools/test/allocation_tracking.cpp:21-90, asserts zero inside process(); here the adapter and libnam_hip.so must make
// none — what the HIP runtime does inside a launch is reported separately).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <dlfcn.h>
#include <new>
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

// ---- allocation accounting (operator new of the whole process resolves here) ----------------------------------------
namespace
{
std::atomic<bool> g_counting{false};
std::atomic<long> g_ours{0}, g_cxx{0}, g_hip{0}, g_other{0};
void note_allocation(void* caller)
{
  if (!g_counting.load(std::memory_order_relaxed))
    return;
  Dl_info info{};
  const char* name = (dladdr(caller, &info) && info.dli_fname) ? info.dli_fname : "";
  if (std::strstr(name, "libnam_hip") || std::strstr(name, "benchmodel"))
    g_ours++;
  else if (std::strstr(name, "libstdc++"))
    g_cxx++; // (std::string / iostream internals: the caller behind them is not visible from here)
  else if (std::strstr(name, "libamdhip") || std::strstr(name, "libhsa") || std::strstr(name, "librocprofiler") || std::strstr(name, "libamd_comgr"))
    g_hip++;
  else
    g_other++;
}
} // namespace
void* operator new(std::size_t n)
{
  note_allocation(__builtin_return_address(0));
  if (void* p = std::malloc(n ? n : 1))
    return p;
  throw std::bad_alloc();
}
void* operator new[](std::size_t n)
{
  note_allocation(__builtin_return_address(0));
  if (void* p = std::malloc(n ? n : 1))
    return p;
  throw std::bad_alloc();
}
void operator delete(void* p) noexcept { std::free(p); }
void operator delete[](void* p) noexcept { std::free(p); }
void operator delete(void* p, std::size_t) noexcept { std::free(p); }
void operator delete[](void* p, std::size_t) noexcept { std::free(p); }

namespace
{
void report(std::vector<double>& us, bool count_allocs)
{
  std::sort(us.begin(), us.end());
  auto pct = [&](double q) { return us[std::min(us.size() - 1, (size_t)(q * (us.size() - 1) + 0.5))]; };
  std::cout << "round trip per buffer (us): min " << us.front() << "  p50 " << pct(0.5) << "  p99 " << pct(0.99) << "  max "
            << us.back() << "\n";
  if (count_allocs)
    std::cout << "allocations in the timed loop: nam_hip+adapter " << g_ours.load() << "  libstdc++ " << g_cxx.load()
              << "  hip-runtime " << g_hip.load() << "  other " << g_other.load() << "\n";
}
} // namespace

int main(int argc, char* argv[])
{
  if (argc < 2)
  {
    std::cerr << "Usage: benchmodel <model_path> [--slim <0..1>] [--no-fast-tanh] [--streams N [--resident | --in-flight D]] [--buffer N] [--count-allocs]\n";
    return 1;
  }
  const char* modelPath = argv[1];
  double slim = -1.0;
  bool fast_tanh = true;
  int streams = 1;
  int bufferSize = AUDIO_BUFFER_SIZE;
  bool count_allocs = false, resident = false;
  const char* kernelName = "auto";
  int inFlight = 0;
  for (int i = 2; i < argc; i++)
  {
    if (!std::strcmp(argv[i], "--count-allocs"))
      count_allocs = true;
    if (!std::strcmp(argv[i], "--slim") && i + 1 < argc)
      slim = std::atof(argv[++i]);
    else if (!std::strcmp(argv[i], "--no-fast-tanh"))
      fast_tanh = false;
    else if (!std::strcmp(argv[i], "--streams") && i + 1 < argc)
      streams = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--buffer") && i + 1 < argc) // frames per process() call (the reference's tool: 64)
      bufferSize = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--in-flight") && i + 1 < argc) // with --streams: host buffers through BatchDSP::submit / wait, D tickets deep
      inFlight = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--resident")) // with --streams: the audio stays in device memory (BatchDSP::process_device)
      resident = true;
    else if (!std::strcmp(argv[i], "--kernel") && i + 1 < argc) // with --streams: auto | generic | a1 | a1_mfma | a1_il | wn_reg (A/B runs)
      kernelName = argv[++i];
  }
  if (fast_tanh)
    nam::activations::Activation::enable_fast_tanh();
  try
  {
    std::cout << "Loading model " << modelPath << "\n";
    const size_t numBuffers = (48000 / bufferSize) * 2;
    if (streams <= 1)
    {
      auto model = nam::get_dsp(modelPath);
      if (slim >= 0.0)
        if (auto* s = dynamic_cast<nam::SlimmableModel*>(model.get()))
          s->SetSlimmableSize(slim);
      model->Reset(model->GetExpectedSampleRate(), bufferSize);
      const int ic = model->NumInputChannels(), oc = model->NumOutputChannels();
      std::vector<std::vector<NAM_SAMPLE>> in(ic, std::vector<NAM_SAMPLE>(bufferSize, 0.0)),
        out(oc, std::vector<NAM_SAMPLE>(bufferSize, 0.0));
      std::vector<NAM_SAMPLE*> inp(ic), outp(oc);
      for (int c = 0; c < ic; c++)
        inp[c] = in[c].data();
      for (int c = 0; c < oc; c++)
        outp[c] = out[c].data();
      std::cout << "Running benchmark\n";
      std::vector<double> us(numBuffers, 0.0);
      for (int i = 0; i < 8; i++) // (first-launch costs stay outside, as the reference's prewarm does for its rings)
        model->process(inp.data(), outp.data(), bufferSize);
      g_counting = count_allocs;
      auto t1 = high_resolution_clock::now();
      for (size_t i = 0; i < numBuffers; i++)
      {
        auto a = high_resolution_clock::now();
        model->process(inp.data(), outp.data(), bufferSize);
        us[i] = duration<double, std::micro>(high_resolution_clock::now() - a).count();
      }
      auto t2 = high_resolution_clock::now();
      g_counting = false;
      duration<double, std::milli> ms = t2 - t1;
      std::cout << duration_cast<milliseconds>(t2 - t1).count() << "ms\n" << ms.count() << "ms\n";
      report(us, count_allocs);
    }
    else
    {
      nam_hip_model* raw = nullptr;
      nam::detail::check(nam_hip_model_load(modelPath, fast_tanh ? 1 : 0, &raw));
      std::shared_ptr<nam_hip_model> m(raw, nam::detail::ModelDeleter());
      nam::BatchDSP batch(m, streams);
      batch.Reset(48000.0, bufferSize);
      if (std::strcmp(kernelName, "auto"))
      {
        const int k = !std::strcmp(kernelName, "generic") ? NAM_HIP_KERNEL_GENERIC
                      : !std::strcmp(kernelName, "a1")    ? NAM_HIP_KERNEL_A1
                      : !std::strcmp(kernelName, "a1_mfma") ? NAM_HIP_KERNEL_A1_MFMA
                      : !std::strcmp(kernelName, "a1_il") ? NAM_HIP_KERNEL_A1_IL
                                                          : NAM_HIP_KERNEL_WN_REG;
        nam::detail::check(nam_hip_batch_set_kernel(batch.GetBatchHandle(), k));
        batch.Reset(48000.0, bufferSize);
      }
      if (slim >= 0.0) // (every stream at that size)
        batch.SetSlimmableSize(nullptr, 0, slim);
      std::cout << "kernel: " << nam_hip_batch_kernel_name_for(batch.GetBatchHandle(), bufferSize) << "\n";
      if (resident)
      {
        // The server shape: the streams' audio lives in device memory (a window of 64 buffers per stream, walked round
        // and round), every buffer is one enqueue-only call, the host waits once per window. Device memory comes from the
        // HIP runtime libnam_hip.so already brought into the process (this tool is plain C++: no HIP headers).
        using malloc_fn = int (*)(void**, size_t);
        using memset_fn = int (*)(void*, int, size_t);
        using free_fn = int (*)(void*);
        void* hip = nullptr;
        for (const char* name : {"libamdhip64.so.7", "libamdhip64.so.6", "libamdhip64.so"}) // the copy that is already loaded
          if (!hip)
            hip = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
        if (!hip)
          hip = dlopen("libamdhip64.so", RTLD_NOW);
        auto hipMalloc_ = hip ? reinterpret_cast<malloc_fn>(dlsym(hip, "hipMalloc")) : nullptr;
        auto hipMemset_ = hip ? reinterpret_cast<memset_fn>(dlsym(hip, "hipMemset")) : nullptr;
        auto hipFree_ = hip ? reinterpret_cast<free_fn>(dlsym(hip, "hipFree")) : nullptr;
        if (!hipMalloc_ || !hipMemset_ || !hipFree_)
          throw std::runtime_error("--resident: the HIP runtime's hipMalloc / hipMemset / hipFree were not found");
        const int kWindow = 64;
        const int64_t stride = (int64_t)kWindow * bufferSize;
        const size_t in_bytes = (size_t)streams * batch.NumInputChannels() * stride * sizeof(float),
                     out_bytes = (size_t)streams * batch.NumOutputChannels() * stride * sizeof(float);
        float *d_in = nullptr, *d_out = nullptr;
        if (hipMalloc_(reinterpret_cast<void**>(&d_in), in_bytes) != 0 || hipMalloc_(reinterpret_cast<void**>(&d_out), out_bytes) != 0)
          throw std::runtime_error("--resident: hipMalloc failed");
        hipMemset_(d_in, 0, in_bytes);
        hipMemset_(d_out, 0, out_bytes);
        batch.synchronize();
        std::cout << "Running benchmark (" << streams << " streams, device-resident buffers)\n";
        auto pass = [&](size_t n) {
          for (size_t i = 0; i < n; i++)
          {
            const int64_t off = (int64_t)(i % kWindow) * bufferSize;
            batch.process_device(d_in + off, d_out + off, bufferSize, stride);
            if ((i + 1) % kWindow == 0)
              batch.flush();
          }
          batch.flush();
          batch.synchronize();
        };
        pass(4 * kWindow); // warm-up (clocks, session start)
        auto t1 = high_resolution_clock::now();
        pass(numBuffers);
        auto t2 = high_resolution_clock::now();
        duration<double, std::milli> ms = t2 - t1;
        std::cout << ms.count() << "ms for 2 s x " << streams << " streams = " << 2000.0 * streams / ms.count() << " x real time ("
                  << ms.count() * 1e3 / (double)numBuffers << " us per buffer, one flush per " << kWindow << " buffers)\n";
        hipFree_(d_in);
        hipFree_(d_out);
        return 0;
      }
      std::vector<float> in((size_t)streams * batch.NumInputChannels() * bufferSize, 0.0f),
        out((size_t)streams * batch.NumOutputChannels() * bufferSize, 0.0f);
      if (inFlight > 0)
      {
        // The feeder shape: a thread hands buffer k in and takes buffer k - D + 1 out, D buffers between the two (an audio
        // server's network thread, a file renderer reading ahead). Each stream's host buffers are distinct per slot.
        const int D = std::min(inFlight, (int)NAM_HIP_PIPE_SLOTS);
        std::vector<std::vector<float>> ins(D, in), outs(D, out);
        std::cout << "Running benchmark (" << streams << " streams, host buffers, " << D << " in flight)\n";
        std::vector<int64_t> tickets(D, -1);
        const bool no_out = std::getenv("NAM_BENCHMODEL_NOOUT") != nullptr; // (developer runs: the waits discard the output)
        for (int w = 0; w < 16; w++)
          batch.wait(batch.submit(ins[0].data(), bufferSize), outs[0].data());
        std::vector<double> us(numBuffers, 0.0);
        double us_wait = 0.0, us_submit = 0.0;
        g_counting = count_allocs;
        auto t1 = high_resolution_clock::now();
        for (size_t i = 0; i < numBuffers; i++)
        {
          auto a = high_resolution_clock::now();
          if (i >= (size_t)D) // the slot's previous ticket
            batch.wait(tickets[i % D], no_out ? nullptr : outs[i % D].data());
          auto m = high_resolution_clock::now();
          tickets[i % D] = batch.submit(ins[i % D].data(), bufferSize);
          auto e = high_resolution_clock::now();
          us[i] = duration<double, std::micro>(e - a).count();
          us_wait += duration<double, std::micro>(m - a).count();
          us_submit += duration<double, std::micro>(e - m).count();
        }
        for (size_t i = numBuffers > (size_t)D ? numBuffers - D : 0; i < numBuffers; i++)
          batch.wait(tickets[i % D], outs[i % D].data());
        auto t2 = high_resolution_clock::now();
        g_counting = false;
        duration<double, std::milli> ms = t2 - t1;
        std::cout << ms.count() << "ms for 2 s x " << streams << " streams = " << 2000.0 * streams / ms.count() << " x real time ("
                  << ms.count() * 1e3 / (double)numBuffers << " us per buffer: wait " << us_wait / (double)numBuffers << ", submit "
                  << us_submit / (double)numBuffers << "; per-buffer figures below: wait for the slot + submit)\n";
        if (std::getenv("NAM_BENCHMODEL_DUMP")) // the first iterations one by one (where do the stalls sit?)
        {
          std::cout << "us per iteration, from iteration 64:";
          for (size_t i = 64; i < std::min<size_t>(numBuffers, 64 + 96); i++)
            std::cout << " " << (int)(us[i] + 0.5);
          std::cout << "\n";
        }
        report(us, count_allocs);
        return 0;
      }
      std::cout << "Running benchmark (" << streams << " streams, host buffers)\n";
      std::vector<double> us(numBuffers, 0.0);
      for (int i = 0; i < 8; i++)
        batch.process_batch(in.data(), out.data(), bufferSize);
      g_counting = count_allocs;
      auto t1 = high_resolution_clock::now();
      for (size_t i = 0; i < numBuffers; i++)
      {
        auto a = high_resolution_clock::now();
        batch.process_batch(in.data(), out.data(), bufferSize);
        us[i] = duration<double, std::micro>(high_resolution_clock::now() - a).count();
      }
      auto t2 = high_resolution_clock::now();
      g_counting = false;
      duration<double, std::milli> ms = t2 - t1;
      std::cout << ms.count() << "ms for 2 s x " << streams << " streams = " << 2000.0 * streams / ms.count()
                << " x real time\n";
      report(us, count_allocs);
    }
  }
  catch (const std::exception& e)
  {
    std::cerr << "Error: " << e.what() << "\n";
    return 1;
  }
  return 0;
}
