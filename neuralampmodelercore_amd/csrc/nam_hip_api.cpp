// nam_hip_api.cpp — the C ABI declared in include/nam_hip.h: the extern "C" entry points (argument checks, the order of
// operations of a call). The runtime behind them: api_launch.cpp, api_session.cpp, api_host_io.cpp (api_internal.h).
// No exception leaves this file.
#include "api_internal.h"

namespace namhip
{
thread_local hipEvent_t tl_session_stop_event = nullptr; // (kernels.h: nam_launch)
namespace api
{
namespace
{
thread_local std::string g_last_error;
}
int fail(int code, const std::string& msg)
{
  g_last_error = msg;
  return code;
}
const std::string& last_error()
{
  return g_last_error;
}
} // namespace api
} // namespace namhip

extern "C" {

const char* nam_hip_last_error(void)
{
  return namhip::api::last_error().c_str();
}

int nam_hip_device_count(int* out_count)
{
  if (!out_count)
    return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_device_count: null argument");
  *out_count = 0;
  NAM_HIP_CHECK(hipGetDeviceCount(out_count));
  return NAM_HIP_OK;
}

int nam_hip_version_support(const char* nam_file_version)
{
  if (!nam_file_version)
    return 0;
  return guarded([&]() { return version_support(nam_file_version); });
}

const char* nam_hip_version(void)
{
  return "nam_hip 0.2.1 gfx950"; // 0.2.1: nam_hip_model_info has_a1_kernel bits 2 and 3 always equal (include/nam_hip.h); 0.2: nam_hip_load_options::struct_size (0.1 callers: the first 16 bytes are read)
}

int nam_hip_model_load(const char* nam_path, int fast_tanh, nam_hip_model** out_model)
{
  if (!nam_path || !out_model)
    return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_model_load: null argument");
  *out_model = nullptr;
  LoadOptions lo;
  lo.fast_tanh = fast_tanh != 0;
  return guarded([&]() { return build_model(load_nam_file(nam_path, lo), out_model); });
}

int nam_hip_model_load_json(const char* json_text, int fast_tanh, nam_hip_model** out_model)
{
  if (!json_text || !out_model)
    return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_model_load_json: null argument");
  *out_model = nullptr;
  LoadOptions lo;
  lo.fast_tanh = fast_tanh != 0;
  return guarded([&]() { return build_model(load_nam_text(json_text, lo), out_model); });
}

int nam_hip_model_load_ex(const char* nam_path, const char* json_text, const nam_hip_load_options* options,
                          nam_hip_model** out_model)
{
  if ((!nam_path) == (!json_text) || !out_model)
    return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_model_load_ex: pass exactly one of nam_path / json_text, and out_model");
  *out_model = nullptr;
  return guarded([&]() {
    LoadOptions lo;
    if (options)
    {
      lo.fast_tanh = options->fast_tanh != 0;
      // (a field behind the original 16 bytes is read only when the caller's struct_size covers it: include/nam_hip.h)
      const bool has_v2 = options->struct_size >= (int32_t)(offsetof(nam_hip_load_options, struct_size) + sizeof(int32_t));
      lo.skip_version_gate = has_v2 && options->version_checked_by_caller != 0;
      if (options->n_luts < 0 || (options->n_luts > 0 && !options->luts))
        throw std::runtime_error("nam_hip_model_load_ex: bad lookup-table list");
      for (int i = 0; i < options->n_luts; i++)
      {
        const nam_hip_lut& l = options->luts[i];
        const std::string name = l.function_name ? l.function_name : "";
        LutSpec ls;
        if (name == "Tanh")
          ls.act_type = ACT_TANH;
        else if (name == "Sigmoid")
          ls.act_type = ACT_SIGMOID;
        else if (name == "SiLU")
          ls.act_type = ACT_SILU;
        else // the reference's message (activations.cpp:209-211)
          throw std::runtime_error("Tried to enable LUT for a function other than Tanh, Sigmoid, or SiLU");
        if (l.n_points < 2 || l.n_points > (1 << 22) || !(l.max_x > l.min_x))
          throw std::runtime_error("nam_hip_model_load_ex: a lookup table needs max > min and 2 <= n_points <= 4194304");
        ls.min_x = l.min_x;
        ls.max_x = l.max_x;
        ls.n_points = l.n_points;
        lo.luts.erase(std::remove_if(lo.luts.begin(), lo.luts.end(), [&](const LutSpec& o) { return o.act_type == ls.act_type; }),
                      lo.luts.end()); // a later enable_lut of the same function replaces the earlier one
        lo.luts.push_back(ls);
      }
    }
    return build_model(nam_path ? load_nam_file(nam_path, lo) : load_nam_text(json_text, lo), out_model);
  });
}

int nam_hip_model_load_parts(const char* version, const char* architecture, const char* config_json, const char* metadata_json,
                             const float* weights, int64_t n_weights, double expected_sample_rate,
                             const nam_hip_load_options* options, nam_hip_model** out_model)
{
  if (!version || !architecture || !config_json || !out_model || n_weights < 0 || (n_weights > 0 && !weights))
    return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_model_load_parts: bad argument");
  *out_model = nullptr;
  // the document get_dsp(const nlohmann::json&) would have been given: one code path for every way in
  std::string doc;
  doc.reserve((size_t)n_weights * 14 + std::strlen(config_json) + 256);
  auto quoted = [&](const char* t) {
    doc += '"';
    for (const char* c = t; *c; c++)
    {
      if (*c == '"' || *c == '\\')
        doc += '\\';
      doc += *c;
    }
    doc += '"';
  };
  doc += "{\"version\":";
  quoted(version);
  doc += ",\"architecture\":";
  quoted(architecture);
  doc += ",\"config\":";
  doc += config_json;
  if (metadata_json && metadata_json[0])
  {
    doc += ",\"metadata\":";
    doc += metadata_json;
  }
  if (expected_sample_rate >= 0.0) // NAM_UNKNOWN_EXPECTED_SAMPLE_RATE = -1.0 (dsp.h:28): no key
  {
    char buf[48];
    std::snprintf(buf, sizeof(buf), ",\"sample_rate\":%.17g", expected_sample_rate);
    doc += buf;
  }
  doc += ",\"weights\":[";
  for (int64_t i = 0; i < n_weights; i++)
  {
    char buf[32];
    const double wv = (double)weights[i];
    if (wv != wv || wv == HUGE_VAL || wv == -HUGE_VAL) // (no JSON form: the spellings json_min.h reads back)
      std::snprintf(buf, sizeof(buf), i ? ",%s" : "%s", wv != wv ? "NaN" : wv > 0 ? "Infinity" : "-Infinity");
    else
      std::snprintf(buf, sizeof(buf), i ? ",%.9g" : "%.9g", wv); // 9 digits: float round trip is exact
    doc += buf;
  }
  doc += "]}";
  return nam_hip_model_load_ex(nullptr, doc.c_str(), options, out_model);
}

int64_t nam_hip_model_get_string(const nam_hip_model* model, int field, char* buf, int64_t capacity)
{
  if (!model || capacity < 0)
    return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_model_get_string: bad argument");
  const ModelSpec& s = *model->spec;
  std::string text;
  switch (field)
  {
    case NAM_HIP_FIELD_VERSION: text = s.version; break;
    case NAM_HIP_FIELD_ARCHITECTURE: text = s.architecture_name; break;
    case NAM_HIP_FIELD_CONFIG_JSON: text = s.config_text; break;
    case NAM_HIP_FIELD_METADATA_JSON: text = s.metadata_text; break;
    case NAM_HIP_FIELD_DESCRIPTION:
      for (size_t i = 0; i < model->plans.size(); i++)
      {
        const Plan& p = model->plans[i];
        text += (i ? " | plan " : "plan ") + std::to_string(i) + ": " + p.describe();
        if (p.arch == ARCH_WAVENET)
          text += std::string(" a1_valu=") + (p.a1.valid ? "1" : "0") + " a1_mfma=" + ((p.a1.valid && p.a1.ws_ok) ? "1" : "0")
                  + " kt_mfma=" + ((p.a1.valid && p.a1.kt_ok) ? "1" : "0") + " kp=" + ((p.a1.valid && p.a1.kp_ok) ? "1" : "0") + " a1_il=" + ((p.a1.valid && p.a1.il_ok) ? "1" : "0")
                  + " a1_p2=" + ((p.a1.valid && p.a1.il_ok && p.a1.p2_ok) ? "1" : "0")
                  + (p.wr.ok ? std::string(" wn_reg=1") : " wn_reg=0 (" + p.wr.why + ")")
                  + (p.wr.jit_failed.empty() ? std::string() : " wn_reg_jit=failed (" + p.wr.jit_failed + ")");
      }
      break;
    default: return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_model_get_string: unknown field");
  }
  if (buf && capacity > 0)
  {
    const size_t n = std::min((size_t)capacity - 1, text.size());
    std::memcpy(buf, text.data(), n);
    buf[n] = 0;
  }
  return (int64_t)text.size();
}

int64_t nam_hip_model_get_weights(const nam_hip_model* model, float* out, int64_t capacity)
{
  if (!model || capacity < 0)
    return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_model_get_weights: bad argument");
  const ModelSpec& s = *model->spec;
  const std::vector<float>* w = s.arch == ARCH_WAVENET ? &s.wavenet.weights : s.arch == ARCH_LSTM ? &s.lstm.weights : nullptr;
  const int64_t n = w ? (int64_t)w->size() : 0;
  if (out && w)
    std::memcpy(out, w->data(), (size_t)std::min(n, capacity) * sizeof(float));
  return n;
}

int nam_hip_sample_rate_from_nam(const char* nam_path, const char* json_text, double* out_sample_rate)
{
  if ((!nam_path) == (!json_text) || !out_sample_rate)
    return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_sample_rate_from_nam: pass exactly one of nam_path / json_text, and out_sample_rate");
  return guarded([&]() {
    std::string text;
    if (nam_path)
    {
      FILE* f = std::fopen(nam_path, "rb");
      if (!f)
        throw FileValidationError(std::string("Could not validate .nam file [") + nam_path + "]: file does not exist.");
      char chunk[65536];
      size_t n;
      while ((n = std::fread(chunk, 1, sizeof(chunk), f)) > 0)
        text.append(chunk, n);
      std::fclose(f);
    }
    else
      text = json_text;
    *out_sample_rate = sample_rate_from_nam_text(text);
    return NAM_HIP_OK;
  });
}

void nam_hip_model_free(nam_hip_model* model)
{
  delete model;
}

int nam_hip_model_get_info(const nam_hip_model* model, nam_hip_model_info* info)
{
  if (!model || !info)
    return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_model_get_info: null argument");
  std::memset(info, 0, sizeof(*info));
  const ModelSpec& s = *model->spec;
  const Plan& p = model->plans[model->full_width];
  info->architecture = s.arch == ARCH_WAVENET ? NAM_HIP_ARCH_WAVENET : s.arch == ARCH_LSTM ? NAM_HIP_ARCH_LSTM : NAM_HIP_ARCH_CONTAINER;
  info->in_channels = s.in_channels();
  info->out_channels = s.out_channels();
  // what GetPrewarmSamples() of the reference object returns: a SlimmableWavenet answers 0 (slimmable.h:66 — its
  // inner WaveNet prewarms inside its own Reset); everything else its own count (container: the active submodel)
  info->prewarm_samples = (s.arch == ARCH_WAVENET && s.wavenet.slimmable) ? 0 : p.prewarm_samples;
  info->expected_sample_rate = s.sample_rate;
  info->has_loudness = s.has_loudness;
  info->has_input_level = s.has_input_level;
  info->has_output_level = s.has_output_level;
  info->is_slimmable = model->slimmable() ? 1 : 0;
  info->loudness = s.loudness;
  info->input_level = s.input_level;
  info->output_level = s.output_level;
  info->num_weights = s.arch == ARCH_WAVENET ? (int64_t)s.wavenet.weights.size()
                      : s.arch == ARCH_LSTM  ? (int64_t)s.lstm.weights.size()
                                             : 0; // a container has no weights of its own
  info->fast_tanh = s.fast_tanh ? 1 : 0;
  info->has_a1_kernel = (p.a1.valid ? 1 : 0) | ((p.a1.valid && (p.a1.ws_ok || p.a1.kt_ok)) ? 2 : 0)
                        | ((p.a1.valid && p.a1.il_ok && p.a1.p2_ok) ? 4 | 8 : 0)
                        | (p.wr.ok ? 16 : 0) | (p.wr.jit_failed.empty() ? 0 : 32);
  info->state_bytes_per_stream = (int64_t)p.state_floats * (int64_t)sizeof(float);
  std::strncpy(info->version, s.version.c_str(), sizeof(info->version) - 1);
  return NAM_HIP_OK;
}

int nam_hip_model_slimmable_breakpoints(const nam_hip_model* model, double* out, int capacity)
{
  if (!model)
    return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_model_slimmable_breakpoints: null model");
  if (!model->slimmable())
    return 0;
  std::vector<double> bp;
  if (model->spec->arch == ARCH_CONTAINER) // every max_value but the last (container.cpp:127-136)
    bp.assign(model->spec->sub_max_value.begin(), model->spec->sub_max_value.end() - 1);
  else
    bp = slimmable_breakpoints(model->spec->wavenet);
  for (int i = 0; i < (int)bp.size() && i < capacity && out; i++)
    out[i] = bp[i];
  return (int)bp.size();
}

int nam_hip_batch_create(const nam_hip_model* model, int device, int n_streams, int max_frames,
                         nam_hip_batch** out_batch)
{
  if (!model || !out_batch || n_streams <= 0 || max_frames <= 0)
    return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_batch_create: bad argument");
  *out_batch = nullptr;
  int count = 0;
  NAM_HIP_CHECK(hipGetDeviceCount(&count));
  if (device < 0 || device >= count)
    return fail(NAM_HIP_ERR_DEVICE, "nam_hip_batch_create: no such HIP device " + std::to_string(device));
  NAM_HIP_CHECK(hipSetDevice(device));
  nam_hip_batch* b = new (std::nothrow) nam_hip_batch();
  if (!b)
    return fail(NAM_HIP_ERR_DEVICE, "nam_hip_batch_create: out of host memory");
  b->model = model;
  b->device = device;
  (void)hipDeviceGetAttribute(&b->n_cus, hipDeviceAttributeMultiprocessorCount, device);
  {
    b->ticket_linger = ticket_linger_from_env();
    if (const char* e = std::getenv("NAM_HIP_BLOCKING_LINGER_US")) // (0: blocking host calls never make a launch linger)
      b->blocking_linger_us = (int)std::min(std::max(std::atol(e), 0l), 100000l);
    if (const char* e = std::getenv("NAM_HIP_MAX_STAGES"))
    {
      const int v = std::atoi(e);
      if (v >= 1)
      {
        b->wr_max_stages = v;
        b->no_pipe = v == 1;
      }
    }
  }
  b->n_streams = n_streams;
  b->max_frames = max_frames;
  // everything that can fail runs inside this lambda; on failure the half-built batch goes through
  // nam_hip_batch_destroy, which frees whatever was already allocated (stream, blobs, state, staging)
  const int rc = [&]() -> int {
    NAM_HIP_CHECK(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    b->groups.resize(model->plans.size());
    for (size_t i = 0; i < model->plans.size(); i++)
    {
      b->groups[i].plan = &model->plans[i];
      const int r = upload_group(b, b->groups[i]);
      if (r != NAM_HIP_OK)
        return r;
    }
    b->stream_width.assign(n_streams, model->full_width);
    WidthGroup& g = b->groups[model->full_width];
    g.streams.resize(n_streams);
    for (int i = 0; i < n_streams; i++)
      g.streams[i] = i;
    const int r = ensure_state(b, g);
    if (r != NAM_HIP_OK)
      return r;
    const size_t in_floats = (size_t)n_streams * model->spec->in_channels() * max_frames;
    const size_t out_floats = (size_t)n_streams * model->spec->out_channels() * max_frames;
    NAM_HIP_CHECK(hipMalloc(&b->d_in, in_floats * sizeof(float)));
    NAM_HIP_CHECK(hipMalloc(&b->d_out, out_floats * sizeof(float)));
    NAM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&b->h_stage), std::max(in_floats, out_floats) * sizeof(float),
                                hipHostMallocDefault));
    NAM_HIP_CHECK(hipStreamSynchronize(b->stream));
    return NAM_HIP_OK;
  }();
  if (rc != NAM_HIP_OK)
  {
    const std::string msg = g_last_error; // destroy must not clobber the reason
    nam_hip_batch_destroy(b);
    g_last_error = msg;
    return rc;
  }
  *out_batch = b;
  return NAM_HIP_OK;
}

void nam_hip_batch_destroy(nam_hip_batch* batch)
{
  if (!batch)
    return;
  (void)hipSetDevice(batch->device);
  (void)quiesce(batch);
  if (const char* e = std::getenv("NAM_HIP_SESSION_STATS"))
    if (e[0] == '1' && batch->ps.n_starts)
      std::fprintf(stderr, "nam_hip session: %llu starts, %llu launches (%llu from a wait / flush), %llu commands stored by the host, %llu through the stream\n",
                   batch->ps.n_starts, batch->ps.n_launches, batch->ps.n_flush_relaunches, batch->ps.n_host_doorbells, batch->ps.n_stream_doorbells);
  if (const char* e = std::getenv("NAM_HIP_SESSION_STATS"))
    if (e[0] == '1' && batch->ps.n_waits)
      std::fprintf(stderr, "nam_hip tickets: %llu waits, %.1f looks at the completion word each\n", batch->ps.n_waits, (double)batch->ps.n_polls / batch->ps.n_waits);
  if (stats_on() && batch->ps.n_waits)
    std::fprintf(stderr, "nam_hip tickets, us per buffer (max): wait for the count %.2f (%.1f), copy out %.2f (%.1f), copy in %.2f (%.1f), commands %.2f (%.1f)\n",
                 batch->ps.t_poll / batch->ps.n_waits, batch->ps.t_poll_max, batch->ps.t_out / batch->ps.n_waits, batch->ps.t_out_max,
                 batch->ps.t_in / batch->ps.n_waits, batch->ps.t_in_max, batch->ps.t_cmd / batch->ps.n_waits, batch->ps.t_cmd_max);
  persist_free(batch);
  for (auto& g : batch->groups)
    free_group(g);
  if (batch->d_in)
    (void)hipFree(batch->d_in);
  if (batch->d_out)
    (void)hipFree(batch->d_out);
  if (batch->h_stage)
    (void)hipHostFree(batch->h_stage);
  if (batch->in_bar)
    (void)hipFree(batch->in_bar);
  if (batch->h_out_map)
    (void)hipHostFree(batch->h_out_map);
  if (batch->pipe_in_bar)
    (void)hipFree(batch->pipe_in_bar);
  if (batch->pipe_h_out_map)
    (void)hipHostFree(batch->pipe_h_out_map);
  if (batch->pipe_h_in)
    (void)hipHostFree(batch->pipe_h_in);
  if (batch->pipe_h_out)
    (void)hipHostFree(batch->pipe_h_out);
  if (batch->pipe_d_in)
    (void)hipFree(batch->pipe_d_in);
  if (batch->pipe_d_out)
    (void)hipFree(batch->pipe_d_out);
  for (PipeSlot& sl : batch->pipe)
    if (sl.done)
      (void)hipEventDestroy(sl.done);
  if (batch->stream)
    (void)hipStreamDestroy(batch->stream);
  delete batch;
}

int nam_hip_batch_reset(nam_hip_batch* batch, int prewarm)
{
  if (!batch)
    return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_batch_reset: null batch");
  NAM_HIP_CHECK(hipSetDevice(batch->device));
  NAM_HIP_CHECK(quiesce(batch)); // launches still running on a caller-supplied stream must not race with the zeroing
  batch->was_reset = true;
  batch->reset_with_prewarm = prewarm != 0;
  for (auto& g : batch->groups)
  {
    if (g.streams.empty())
      continue;
    const int rc = reset_streams(batch, g, g.d_map, (int)g.streams.size(), prewarm != 0, g.streams.front());
    if (rc != NAM_HIP_OK)
      return rc;
  }
  NAM_HIP_CHECK(hipStreamSynchronize(batch->stream));
  if (batch->ps.enabled && persist_eligible(batch)) // (a session's window-independent resources: see nam_hip_batch_set_persistent)
  {
    const int rc = persist_prepare(batch);
    if (rc != NAM_HIP_OK)
      return rc;
  }
  return NAM_HIP_OK;
}

int nam_hip_batch_set_slimmable_size(nam_hip_batch* batch, const int* stream_ids, int n_ids, double ratio)
{
  if (!batch)
    return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_batch_set_slimmable_size: null batch");
  const nam_hip_model* m = batch->model;
  if (!m->slimmable())
    return NAM_HIP_OK; // not a SlimmableModel: the reference's dynamic_cast fails and callers skip
  NAM_HIP_CHECK(hipSetDevice(batch->device));
  const int w = m->width_for_ratio(ratio);
  if (w < 0)
    return fail(NAM_HIP_ERR_MODEL, "nam_hip_batch_set_slimmable_size: no plan for this ratio");
  std::vector<int> ids;
  if (!stream_ids)
  {
    ids.resize(batch->n_streams);
    for (int i = 0; i < batch->n_streams; i++)
      ids[i] = i;
  }
  else
  {
    for (int i = 0; i < n_ids; i++)
    {
      if (stream_ids[i] < 0 || stream_ids[i] >= batch->n_streams)
        return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_batch_set_slimmable_size: stream id out of range");
      ids.push_back(stream_ids[i]);
    }
  }
  std::sort(ids.begin(), ids.end());
  ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
  std::vector<int> moved;
  for (int s : ids)
    if (batch->stream_width[s] != w) // same width: no-op (slimmable.cpp:459-464)
      moved.push_back(s);
  if (moved.empty())
    return NAM_HIP_OK;
  NAM_HIP_CHECK(quiesce(batch));
  for (int s : moved)
  {
    auto& old = batch->groups[batch->stream_width[s]].streams;
    old.erase(std::remove(old.begin(), old.end(), s), old.end());
    batch->stream_width[s] = w;
    batch->groups[w].streams.push_back(s);
  }
  std::sort(batch->groups[w].streams.begin(), batch->groups[w].streams.end());
  int rc = ensure_state(batch, batch->groups[w]);
  if (rc != NAM_HIP_OK)
    return rc;
  for (auto& g : batch->groups)
  {
    rc = refresh_map(batch, g);
    if (rc != NAM_HIP_OK)
      return rc;
  }
  // fresh sub-model state for the streams that moved: Reset (+ prewarm) as in slimmable.cpp:433-447
  int* d_moved = nullptr;
  NAM_HIP_CHECK(hipMalloc(&d_moved, moved.size() * sizeof(int)));
  NAM_HIP_CHECK(hipMemcpy(d_moved, moved.data(), moved.size() * sizeof(int), hipMemcpyHostToDevice));
  rc = reset_streams(batch, batch->groups[w], d_moved, (int)moved.size(), batch->was_reset && batch->reset_with_prewarm, moved.front());
  hipError_t e = hipStreamSynchronize(batch->stream);
  (void)hipFree(d_moved);
  if (rc != NAM_HIP_OK)
    return rc;
  NAM_HIP_CHECK(e);
  return NAM_HIP_OK;
}

int nam_hip_batch_process_device(nam_hip_batch* batch, const float* d_in, float* d_out, int n_frames,
                                 int64_t frame_stride, void* hip_stream)
{
  if (!batch || !d_in || !d_out || n_frames < 0)
    return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_batch_process_device: bad argument");
  if (frame_stride < n_frames)
    return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_batch_process_device: frame_stride < n_frames");
  NAM_HIP_CHECK(hipSetDevice(batch->device));
  hipStream_t s = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : batch->stream;
  if (hip_stream)
    batch->last_ext_stream = s;
  if (batch->ps.enabled && persist_eligible(batch))
  {
    // fresh state has the layout any kernel family writes; anything else must already be the session kernel's
    for (auto& g0 : batch->groups)
      if (!g0.streams.empty() && g0.plan->arch == ARCH_WAVENET && g0.state_family >= 0
          && g0.state_family != persist_family(batch, g0))
        return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "persistent mode: the state was written in another kernel family's layout; reset first");
    const int rc = persist_submit(batch, d_in, d_out, n_frames, (long)frame_stride, s);
    if (rc <= 0)
      return rc; // submitted (0) or failed (< 0)
  }
  if (batch->ps.active) // this call is not a 64-frame buffer of the session: the resident launch hands the state back first
  {
    const int rc = persist_stop(batch);
    if (rc != NAM_HIP_OK)
      return rc;
  }
  {
    // every group on nam_wn_reg_kernel: one launch for the whole (mixed-width) batch
    const WrGroupList gs = wr_groups(batch);
    if (gs.n > 1)
      return n_frames > 0 ? launch_wr_all(batch, gs, d_in, d_out, n_frames, (long)frame_stride, s) : NAM_HIP_OK;
  }
  for (auto& g : batch->groups)
  {
    if (g.streams.empty())
      continue;
    const int rc = launch_group(batch, g, g.d_map, (int)g.streams.size(), d_in, d_out, n_frames, (long)frame_stride, s);
    if (rc != NAM_HIP_OK)
      return rc;
  }
  return NAM_HIP_OK;
}

int nam_hip_batch_process_f32(nam_hip_batch* batch, const float* in, float* out, int n_frames)
{
  if (!batch || !in || !out || n_frames < 0)
    return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_batch_process_f32: bad argument");
  if (n_frames > batch->max_frames)
    return fail(NAM_HIP_ERR_TOO_MANY_FRAMES, "nam_hip_batch_process_f32: n_frames exceeds max_frames");
  if (n_frames == 0)
    return NAM_HIP_OK;
  NAM_HIP_CHECK(hipSetDevice(batch->device));
  const size_t in_bytes = (size_t)batch->n_streams * batch->model->spec->in_channels() * n_frames * sizeof(float);
  const size_t out_bytes = (size_t)batch->n_streams * batch->model->spec->out_channels() * n_frames * sizeof(float);
  {
    const int rc = process_host_mapped(batch, in, nullptr, out, nullptr, n_frames);
    if (rc <= 0)
      return rc; // done through the session (0) or failed (< 0); 1 = not applicable: the copying path below
  }
  NAM_HIP_CHECK(hipMemcpyAsync(batch->d_in, in, in_bytes, hipMemcpyHostToDevice, batch->stream));
  const int rc = nam_hip_batch_process_device(batch, batch->d_in, batch->d_out, n_frames, n_frames, nullptr);
  if (rc != NAM_HIP_OK)
    return rc;
  if (batch->ps.active) // persistent mode: the buffer is done when every workgroup has published its count
  {
    const int rw = persist_flush(batch, batch->stream);
    if (rw != NAM_HIP_OK)
      return rw;
  }
  NAM_HIP_CHECK(hipMemcpyAsync(out, batch->d_out, out_bytes, hipMemcpyDeviceToHost, batch->stream));
  NAM_HIP_CHECK(hipStreamSynchronize(batch->stream));
  return NAM_HIP_OK;
}

int nam_hip_batch_submit_f32(nam_hip_batch* batch, const float* in, int n_frames, int64_t* out_ticket)
{
  if (!batch || !in || !out_ticket || n_frames <= 0)
    return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_batch_submit_f32: bad argument");
  if (n_frames > batch->max_frames)
    return fail(NAM_HIP_ERR_TOO_MANY_FRAMES, "nam_hip_batch_submit_f32: n_frames exceeds max_frames");
  const int slot = (int)(batch->pipe_next % NAM_HIP_PIPE_SLOTS);
  PipeSlot& sl = batch->pipe[slot];
  if (sl.in_flight)
    return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_batch_submit_f32: " + std::to_string(NAM_HIP_PIPE_SLOTS) + " buffers are in flight; wait for ticket "
                                                + std::to_string(sl.ticket) + " first");
  return guarded([&]() -> int {
    NAM_HIP_CHECK(hipSetDevice(batch->device));
    const int rc = pipe_submit(batch, in, n_frames, sl, slot);
    if (rc != NAM_HIP_OK)
      return rc;
    sl.ticket = batch->pipe_next++;
    sl.in_flight = true;
    *out_ticket = sl.ticket;
    return NAM_HIP_OK;
  });
}

int nam_hip_batch_wait_f32(nam_hip_batch* batch, int64_t ticket, float* out)
{
  if (!batch || ticket < 0)
    return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_batch_wait_f32: bad argument");
  const int slot = (int)(ticket % NAM_HIP_PIPE_SLOTS);
  PipeSlot& sl = batch->pipe[slot];
  if (!sl.in_flight || sl.ticket != ticket)
    return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_batch_wait_f32: ticket " + std::to_string(ticket) + " is not in flight (never issued, or waited for already)");
  return guarded([&]() -> int {
    NAM_HIP_CHECK(hipSetDevice(batch->device));
    return pipe_wait(batch, sl, slot, out);
  });
}

// NAM_SAMPLE = double callers (NAM/dsp.h:18-22): the casts of the blocking _f64 form (in: model.cpp:817, out: :896) around the float32 ticket
int nam_hip_batch_submit_f64(nam_hip_batch* batch, const double* in, int n_frames, int64_t* out_ticket)
{
  if (!batch || !in || !out_ticket || n_frames <= 0)
    return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_batch_submit_f64: bad argument");
  if (n_frames > batch->max_frames)
    return fail(NAM_HIP_ERR_TOO_MANY_FRAMES, "nam_hip_batch_submit_f64: n_frames exceeds max_frames");
  return guarded([&]() -> int {
    const size_t n_in = (size_t)batch->n_streams * batch->model->spec->in_channels() * n_frames;
    const size_t cap = (size_t)batch->n_streams * std::max(batch->model->spec->in_channels(), batch->model->spec->out_channels()) * batch->max_frames;
    if (batch->pipe_cvt.size() < cap)
      batch->pipe_cvt.resize(cap);
    for (size_t i = 0; i < n_in; i++)
      batch->pipe_cvt[i] = (float)in[i];
    return nam_hip_batch_submit_f32(batch, batch->pipe_cvt.data(), n_frames, out_ticket);
  });
}

int nam_hip_batch_wait_f64(nam_hip_batch* batch, int64_t ticket, double* out)
{
  if (!batch || ticket < 0)
    return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_batch_wait_f64: bad argument");
  return guarded([&]() -> int {
    const PipeSlot& sl = batch->pipe[(int)(ticket % NAM_HIP_PIPE_SLOTS)];
    const int n_frames = sl.n_frames; // (of the ticket, if it is the one in flight: the f32 form checks that)
    const size_t cap = (size_t)batch->n_streams * std::max(batch->model->spec->in_channels(), batch->model->spec->out_channels()) * batch->max_frames;
    if (batch->pipe_cvt.size() < cap)
      batch->pipe_cvt.resize(cap);
    const int rc = nam_hip_batch_wait_f32(batch, ticket, out ? batch->pipe_cvt.data() : nullptr);
    if (rc != NAM_HIP_OK || !out)
      return rc;
    const size_t n_out = (size_t)batch->n_streams * batch->model->spec->out_channels() * n_frames;
    for (size_t i = 0; i < n_out; i++)
      out[i] = (double)batch->pipe_cvt[i];
    return NAM_HIP_OK;
  });
}

int nam_hip_batch_render_f32(nam_hip_batch* batch, const float* const* in, float* const* out, const int64_t* n_frames)
{
  if (!batch || !in || !out || !n_frames)
    return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_batch_render_f32: bad argument");
  const int N = batch->n_streams;
  const int ic = batch->model->spec->in_channels(), oc = batch->model->spec->out_channels();
  int64_t T = 0;
  for (int s = 0; s < N; s++)
  {
    if (n_frames[s] < 0 || (n_frames[s] > 0 && (!in[s] || !out[s])))
      return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_batch_render_f32: bad signal pointer / length");
    T = std::max(T, n_frames[s]);
  }
  if (T == 0)
    return NAM_HIP_OK;
  if (T > (int64_t)1 << 30)
    return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_batch_render_f32: signal too long for one launch");
  NAM_HIP_CHECK(hipSetDevice(batch->device));
  // device-resident planar audio [stream][channel][T]; rows of shorter signals are zero-padded
  float *d_in = nullptr, *d_out = nullptr;
  const size_t in_bytes = (size_t)N * ic * T * sizeof(float), out_bytes = (size_t)N * oc * T * sizeof(float);
  NAM_HIP_CHECK(hipMalloc(&d_in, in_bytes));
  hipError_t e = hipMalloc(&d_out, out_bytes);
  int rc = NAM_HIP_OK;
  if (e == hipSuccess)
    e = hipMemsetAsync(d_in, 0, in_bytes, batch->stream);
  for (int s = 0; s < N && e == hipSuccess; s++)
    if (n_frames[s] > 0)
      e = hipMemcpy2DAsync(d_in + (size_t)s * ic * T, (size_t)T * sizeof(float), in[s], (size_t)n_frames[s] * sizeof(float),
                           (size_t)n_frames[s] * sizeof(float), ic, hipMemcpyHostToDevice, batch->stream);
  if (e == hipSuccess)
    rc = nam_hip_batch_process_device(batch, d_in, d_out, (int)T, T, nullptr);
  if (e == hipSuccess && rc == NAM_HIP_OK && batch->ps.active) // (a short render went through the session: it ends here —
    rc = persist_stop(batch);                                   // the window is about to be freed)
  for (int s = 0; s < N && e == hipSuccess && rc == NAM_HIP_OK; s++)
    if (n_frames[s] > 0)
      e = hipMemcpy2DAsync(out[s], (size_t)n_frames[s] * sizeof(float), d_out + (size_t)s * oc * T, (size_t)T * sizeof(float),
                           (size_t)n_frames[s] * sizeof(float), oc, hipMemcpyDeviceToHost, batch->stream);
  const hipError_t es = hipStreamSynchronize(batch->stream);
  (void)hipFree(d_in);
  if (d_out)
    (void)hipFree(d_out);
  if (rc != NAM_HIP_OK)
    return rc;
  NAM_HIP_CHECK(e);
  NAM_HIP_CHECK(es);
  return NAM_HIP_OK;
}

int nam_hip_batch_process_f64(nam_hip_batch* batch, const double* in, double* out, int n_frames)
{
  if (!batch || !in || !out || n_frames < 0)
    return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_batch_process_f64: bad argument");
  if (n_frames > batch->max_frames)
    return fail(NAM_HIP_ERR_TOO_MANY_FRAMES, "nam_hip_batch_process_f64: n_frames exceeds max_frames");
  if (n_frames == 0)
    return NAM_HIP_OK;
  NAM_HIP_CHECK(hipSetDevice(batch->device));
  const size_t n_in = (size_t)batch->n_streams * batch->model->spec->in_channels() * n_frames;
  const size_t n_out = (size_t)batch->n_streams * batch->model->spec->out_channels() * n_frames;
  {
    const int rc = process_host_mapped(batch, nullptr, in, nullptr, out, n_frames);
    if (rc <= 0)
      return rc;
  }
  // double -> float exactly as _set_condition_array does (NAM/wavenet/model.cpp:817)
  for (size_t i = 0; i < n_in; i++)
    batch->h_stage[i] = (float)in[i];
  NAM_HIP_CHECK(hipMemcpyAsync(batch->d_in, batch->h_stage, n_in * sizeof(float), hipMemcpyHostToDevice, batch->stream));
  const int rc = nam_hip_batch_process_device(batch, batch->d_in, batch->d_out, n_frames, n_frames, nullptr);
  if (rc != NAM_HIP_OK)
    return rc;
  if (batch->ps.active)
  {
    const int rw = persist_flush(batch, batch->stream);
    if (rw != NAM_HIP_OK)
      return rw;
  }
  NAM_HIP_CHECK(
    hipMemcpyAsync(batch->h_stage, batch->d_out, n_out * sizeof(float), hipMemcpyDeviceToHost, batch->stream));
  NAM_HIP_CHECK(hipStreamSynchronize(batch->stream));
  for (size_t i = 0; i < n_out; i++)
    out[i] = (double)batch->h_stage[i];
  return NAM_HIP_OK;
}

int nam_hip_batch_synchronize(nam_hip_batch* batch)
{
  if (!batch)
    return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_batch_synchronize: null batch");
  NAM_HIP_CHECK(hipSetDevice(batch->device));
  // "nothing of this batch is running on the device any more": a persistent session ends here (its resident launch
  // would otherwise keep a device-wide hipDeviceSynchronize waiting until it expires)
  NAM_HIP_CHECK(quiesce(batch));
  return NAM_HIP_OK;
}

int nam_hip_batch_set_persistent(nam_hip_batch* batch, int enable)
{
  if (!batch)
    return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_batch_set_persistent: null batch");
  NAM_HIP_CHECK(hipSetDevice(batch->device));
  if (!enable && batch->ps.active)
  {
    const int rc = persist_stop(batch);
    if (rc != NAM_HIP_OK)
      return rc;
  }
  batch->ps.enabled = enable != 0; // (persist_eligible / persist_kind look at it)
  const bool eligible = batch->ps.enabled && persist_eligible(batch);
  if (eligible)
  {
    // this call and Reset are the non-real-time side of the contract (NAM/dsp.h:163): the session's ring, words, stream and the
    // blocking entry points' host windows are allocated here, so that no process call ever allocates (the first used to: ~7 ms)
    const int rc = persist_prepare(batch);
    if (rc != NAM_HIP_OK)
      return rc; // (the mode is off again: persist_prepare released what it had and cleared `enabled`)
    // the blocking entry points' host windows too, when they are small (a batch of long buffers that only ever runs
    // device-resident audio must not pin hundreds of MB for calls it never makes: such a batch allocates them at its first
    // host-buffer call, if it makes one)
    (void)host_windows(batch, 1, batch->in_bar, batch->h_out_map, batch->d_out_map, batch->map_failed, /*prealloc=*/true);
  }
  return eligible ? 1 : 0;
}

int nam_hip_batch_flush(nam_hip_batch* batch, void* hip_stream)
{
  if (!batch)
    return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_batch_flush: null batch");
  NAM_HIP_CHECK(hipSetDevice(batch->device));
  if (!batch->ps.active)
    return NAM_HIP_OK;
  // the doorbells were enqueued on the caller's stream: they are only guaranteed to have been rung once it has drained
  hipStream_t s = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : batch->stream;
  return persist_flush(batch, s);
}

int nam_hip_batch_set_kernel(nam_hip_batch* batch, int kernel)
{
  if (!batch || kernel < NAM_HIP_KERNEL_AUTO || kernel > NAM_HIP_KERNEL_WN_REG)
    return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_batch_set_kernel: bad argument");
  NAM_HIP_CHECK(hipSetDevice(batch->device)); // (ending a session may relaunch: the launchers configure the current device)
  if (batch->ps.active)
  {
    const int rc = persist_stop(batch);
    if (rc != NAM_HIP_OK)
      return rc;
  }
  bool all_lstm = true;
  for (const auto& g : batch->groups)
    all_lstm = all_lstm && g.plan->arch == ARCH_LSTM;
  if (all_lstm)
  {
    // LSTM batches: AUTO (gate-row kernel for small cells, else matrix cores), GENERIC (lanes = streams),
    // A1_MFMA (force the matrix-core kernel)
    if (kernel == NAM_HIP_KERNEL_A1 || kernel == NAM_HIP_KERNEL_A1_IL || kernel == NAM_HIP_KERNEL_WN_REG)
      return fail(NAM_HIP_ERR_UNSUPPORTED, "nam_hip_batch_set_kernel: WaveNet kernels cannot run an LSTM");
    batch->kernel = kernel;
    return NAM_HIP_OK;
  }
  if (kernel == NAM_HIP_KERNEL_WN_REG)
  {
    const Plan& full = *batch->groups[batch->model->full_width].plan;
    if (full.arch != ARCH_WAVENET || !full.wr.ok)
      return fail(NAM_HIP_ERR_UNSUPPORTED, "nam_hip_batch_set_kernel: the register-resident WaveNet kernel cannot run this model ("
                                             + full.wr.why + ")");
  }
  else if (kernel >= NAM_HIP_KERNEL_A1)
  {
    // every submodel must have an A1 plan; the MFMA kernels must exist for the full-width submodel (narrower
    // submodels of a container fall back to the VALU kernel: A2-Lite has 3 channels)
    for (const auto& g : batch->groups)
      if (g.plan->arch != ARCH_WAVENET || !g.plan->a1.valid)
        return fail(NAM_HIP_ERR_UNSUPPORTED, "nam_hip_batch_set_kernel: the A1 kernels cannot run this model");
    const A1Plan& full = batch->groups[batch->model->full_width].plan->a1;
    if (kernel == NAM_HIP_KERNEL_A1_MFMA && !full.ws_ok && !full.kt_ok)
      return fail(NAM_HIP_ERR_UNSUPPORTED, "nam_hip_batch_set_kernel: the A1 MFMA kernel cannot run this model");
    if (kernel == NAM_HIP_KERNEL_A1_IL && !full.il_ok)
      return fail(NAM_HIP_ERR_UNSUPPORTED, "nam_hip_batch_set_kernel: the interleaved-frame MFMA kernel cannot run this model");
  }
  // a change of ring layout (channel-padded models: op program <-> A1 kernels) needs freshly reset state
  const int prev = batch->kernel;
  batch->kernel = kernel;
  for (const auto& g : batch->groups)
    if (g.plan->arch == ARCH_WAVENET && !g.streams.empty() && g.state_family >= 0
        && state_family_of(*g.plan, pick_kernel(batch, g)) != g.state_family)
    {
      batch->kernel = prev;
      return fail(NAM_HIP_ERR_INVALID_ARGUMENT,
                  "nam_hip_batch_set_kernel: the state was written in another kernel family's layout (op program rings / "
                  "zero-padded A1 rings / nam_wn_reg_kernel histories); call nam_hip_batch_reset(batch, 0) first, then "
                  "switch, then reset / prewarm");
    }
  return NAM_HIP_OK;
}

int nam_hip_batch_get_kernel(const nam_hip_batch* batch)
{
  if (!batch)
    return NAM_HIP_ERR_INVALID_ARGUMENT;
  const WidthGroup& g = batch->groups[batch->model->full_width];
  if (g.plan->arch != ARCH_WAVENET)
    return NAM_HIP_KERNEL_GENERIC;
  return pick_kernel(batch, g);
}

// Developer tool (not part of the drop-in surface): run `n_frames` of silence through the MFMA kernel's
// profiling instantiation and copy out its per-wavefront counters (include/nam_hip.h).
int nam_hip_batch_debug_timeline(nam_hip_batch* batch, int n_frames, long long* out_stamps)
{
  if (!batch || !out_stamps)
    return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_hip_batch_debug_timeline: bad argument");
  NAM_HIP_CHECK(hipSetDevice(batch->device));
  const size_t bytes = 96 * 8 * sizeof(long long);
  NAM_HIP_CHECK(hipMalloc(&batch->dbg, bytes));
  NAM_HIP_CHECK(hipMemset(batch->dbg, 0, bytes));
  NAM_HIP_CHECK(hipDeviceSynchronize()); // (the fill runs on the null stream, the launch on the batch's non-blocking one)
  int rc = NAM_HIP_OK;
  for (auto& g : batch->groups)
    if (!g.streams.empty() && rc == NAM_HIP_OK)
      rc = launch_group(batch, g, g.d_map, (int)g.streams.size(), nullptr, nullptr, n_frames, 0, batch->stream);
  hipError_t e = hipStreamSynchronize(batch->stream);
  if (e == hipSuccess)
    e = hipMemcpy(out_stamps, batch->dbg, bytes, hipMemcpyDeviceToHost);
  (void)hipFree(batch->dbg);
  batch->dbg = nullptr;
  if (rc != NAM_HIP_OK)
    return rc;
  NAM_HIP_CHECK(e);
  return NAM_HIP_OK;
}

const char* nam_hip_batch_kernel_name(const nam_hip_batch* batch)
{
  return nam_hip_batch_kernel_name_for(batch, kBlock);
}

const char* nam_hip_batch_kernel_name_for(const nam_hip_batch* batch, int n_frames)
{
  if (!batch)
    return "";
  // the group with the most streams (a uniform batch has exactly one populated group)
  const WidthGroup* best = &batch->groups[batch->model->full_width];
  for (const auto& g : batch->groups)
    if (g.streams.size() > best->streams.size())
      best = &g;
  return group_kernel_name(batch, *best, n_frames > 0 ? n_frames : kBlock);
}

int nam_hip_batch_n_streams(const nam_hip_batch* batch)
{
  return batch ? batch->n_streams : NAM_HIP_ERR_INVALID_ARGUMENT;
}

} // extern "C"
