// ref_wrapper.cpp — C entry points around the UNMODIFIED reference sources (compiled from /root/reference by
// oracle/Makefile.ref against oracle/eigen_shim). TEST INFRASTRUCTURE ONLY: used to validate oracle/nam_oracle.c
// against the reference's own control flow (weight binding, ring buffers, head accumulation, prewarm, container
// and slimmable dispatch) and as the "reference" kind of CPU baseline. Arithmetic inside Eigen expressions is the
// shim's (plain ordered sums), so agreement with the oracle is expected to a few ulp, not bit for bit.
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "NAM/activations.h"
#include "NAM/dsp.h"
#include "NAM/get_dsp.h"
#include "NAM/slimmable.h"

namespace
{
struct Handle
{
  std::unique_ptr<nam::DSP> dsp;
  bool fast_tanh = false;
  std::vector<std::vector<NAM_SAMPLE>> in, out;
  std::vector<NAM_SAMPLE*> inp, outp;
};
void apply_mode(const Handle* h)
{
  // the reference keeps this as a process-global (NAM/activations.cpp:168-187)
  if (h->fast_tanh)
    nam::activations::Activation::enable_fast_tanh();
  else
    nam::activations::Activation::disable_fast_tanh();
}
thread_local std::string g_last; // what the last failing ref_reset / ref_process threw (ref_last_error)
void set_err(char* err, int n, const std::string& s)
{
  if (err && n > 0)
  {
    std::strncpy(err, s.c_str(), (size_t)n - 1);
    err[n - 1] = 0;
  }
}
} // namespace

extern "C"
{
void* ref_load(const char* path, int fast_tanh, char* err, int errlen)
{
  try
  {
    auto h = std::make_unique<Handle>();
    h->fast_tanh = fast_tanh != 0;
    apply_mode(h.get());
    h->dsp = nam::get_dsp(std::filesystem::path(path));
    if (!h->dsp)
    {
      set_err(err, errlen, "get_dsp returned null");
      return nullptr;
    }
    return h.release();
  }
  catch (const std::exception& e)
  {
    set_err(err, errlen, e.what());
    return nullptr;
  }
}
// nam::activations::Activation::enable_lut(name, min, max, n) around get_dsp (NAM/activations.cpp:189-212): the
// registry entry is what a model binds at construction, so the table stays with the loaded model
void* ref_load_lut(const char* path, int fast_tanh, const char* lut_name, float lut_min, float lut_max, int lut_n,
                   char* err, int errlen)
{
  if (fast_tanh)
  {
    // enable_lut("Tanh") and enable_fast_tanh share one backup slot (activations.cpp:169-199): combining them loses the
    // libm tanh for the rest of the process. The wrapper is shared by every test, so it refuses.
    set_err(err, errlen, "ref_load_lut: not with fast_tanh (the reference's shared tanh_bak would be clobbered)");
    return nullptr;
  }
  try
  {
    nam::activations::Activation::disable_fast_tanh();
    nam::activations::Activation::enable_lut(lut_name, lut_min, lut_max, (std::size_t)lut_n);
  }
  catch (const std::exception& e)
  {
    set_err(err, errlen, e.what());
    return nullptr;
  }
  void* h = ref_load(path, fast_tanh, err, errlen);
  nam::activations::Activation::disable_lut(lut_name);
  return h;
}
void ref_free(void* p)
{
  delete static_cast<Handle*>(p);
}
int ref_in_channels(void* p)
{
  return static_cast<Handle*>(p)->dsp->NumInputChannels();
}
int ref_out_channels(void* p)
{
  return static_cast<Handle*>(p)->dsp->NumOutputChannels();
}
int ref_prewarm_samples(void* p)
{
  return static_cast<Handle*>(p)->dsp->GetPrewarmSamples();
}
double ref_expected_sample_rate(void* p)
{
  return static_cast<Handle*>(p)->dsp->GetExpectedSampleRate();
}
int ref_set_slimmable(void* p, double v)
{
  auto* h = static_cast<Handle*>(p);
  auto* s = dynamic_cast<nam::SlimmableModel*>(h->dsp.get());
  if (!s)
    return -1;
  apply_mode(h);
  s->SetSlimmableSize(v);
  return 0;
}
int ref_reset(void* p, double sample_rate, int max_buffer)
{
  auto* h = static_cast<Handle*>(p);
  try
  {
    apply_mode(h);
    h->dsp->Reset(sample_rate, max_buffer);
    const int ic = h->dsp->NumInputChannels(), oc = h->dsp->NumOutputChannels();
    h->in.assign(ic, std::vector<NAM_SAMPLE>((size_t)max_buffer));
    h->out.assign(oc, std::vector<NAM_SAMPLE>((size_t)max_buffer));
    h->inp.resize(ic);
    h->outp.resize(oc);
    for (int c = 0; c < ic; c++)
      h->inp[c] = h->in[c].data();
    for (int c = 0; c < oc; c++)
      h->outp[c] = h->out[c].data();
    return 0;
  }
  catch (const std::exception& e)
  {
    g_last = e.what();
    return -1;
  }
  catch (...)
  {
    g_last = "unknown exception";
    return -1;
  }
}
const char* ref_last_error(void)
{
  return g_last.c_str();
}
// in: planar float32 [in_channels][n_frames], out: [out_channels][n_frames]; fed in `block`-frame process() calls
int ref_process(void* p, const float* in, float* out, long n_frames, int block)
{
  auto* h = static_cast<Handle*>(p);
  if (h->in.empty() || block > (int)h->in[0].size())
    return -1;
  apply_mode(h);
  const int ic = (int)h->in.size(), oc = (int)h->out.size();
  for (long s = 0; s < n_frames; s += block)
  {
    const int n = (int)std::min<long>(block, n_frames - s);
    for (int c = 0; c < ic; c++)
      for (int i = 0; i < n; i++)
        h->in[c][i] = (NAM_SAMPLE)in[(size_t)c * n_frames + s + i];
    h->dsp->process(h->inp.data(), h->outp.data(), n);
    for (int c = 0; c < oc; c++)
      for (int i = 0; i < n; i++)
        out[(size_t)c * n_frames + s + i] = (float)h->out[c][i];
  }
  return 0;
}
}
