// api_host_io.cpp — host buffers in, host buffers out: the windows both sides can reach, blocking calls through a session,
// the ticketed submit / wait. See api_internal.h.
#include "api_internal.h"

namespace namhip
{
namespace api
{

// The blocking entry points inside a persistent session: the kernel reads the buffer from and writes it to HOST-MAPPED
// memory (float32 rows [stream][channel][max_frames]); in_f32 / in_f64 and out_f32 / out_f64: exactly one of each.
// The host-mapped windows of the session's blocking (`slots` = 1: nam_hip_batch::in_bar, h_out_map) or ticketed
// (NAM_HIP_PIPE_SLOTS: pipe_in_bar, pipe_h_out_map) entry points. false: no such memory here (the copying path serves the call).
bool host_windows(nam_hip_batch* b, int slots, float*& in_bar, float*& h_out_map, float*& d_out_map, bool& failed, bool prealloc)
{
  if (failed)
    return false;
  if (in_bar)
    return true;
  const int ic = b->model->spec->in_channels(), oc = b->model->spec->out_channels();
  const size_t pitch = (size_t)b->n_streams * std::max(ic, oc) * b->max_frames; // (one slot; the same for both windows: a command carries ONE offset)
  if (prealloc && pitch * (size_t)slots * sizeof(float) > ((size_t)64 << 20))
    return false; // (ahead of any host-buffer call: only when cheap — nothing decided, nothing said)
  if (pitch * (size_t)slots > (size_t)0x1fff0000)
  {
    // the kernels address a session's window through one 2 GB buffer descriptor: windows beyond it would make every buffer a
    // session of its own (stop, start, launch). Said once; the copying path (staging + launches on the batch's stream) serves
    // such batches
    std::fprintf(stderr, "nam_hip: %d streams x %d frames x %d host-buffer slots exceed the 2 GB session window: host buffers of this "
                         "batch go through staging copies instead of the mapped windows (smaller max_frames or fewer streams per batch avoid this)\n",
                 b->n_streams, b->max_frames, slots);
    failed = true;
    return false;
  }
  if (hipExtMallocWithFlags(reinterpret_cast<void**>(&in_bar), pitch * slots * sizeof(float), hipDeviceMallocFinegrained) != hipSuccess)
  {
    (void)hipGetLastError();
    in_bar = nullptr;
    failed = true; // no host-writable device memory here
    return false;
  }
  // all three or none: a later call must not find the input window without the output window (it would submit commands
  // with a null output base and copy from a null mapping)
  if (hipHostMalloc(reinterpret_cast<void**>(&h_out_map), pitch * slots * sizeof(float), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess
      || hipHostGetDevicePointer(reinterpret_cast<void**>(&d_out_map), h_out_map, 0) != hipSuccess)
  {
    (void)hipGetLastError();
    if (h_out_map)
      (void)hipHostFree(h_out_map);
    (void)hipFree(in_bar);
    in_bar = nullptr;
    h_out_map = nullptr;
    d_out_map = nullptr;
    failed = true; // the copying path takes over
    return false;
  }
  return true;
}


bool host_mapped_applies(nam_hip_batch* b, int n_frames)
{
  if (!b->ps.enabled || n_frames % kBlock != 0 || n_frames > kPersistMaxFrames || !persist_eligible(b))
    return false;
  for (auto& g0 : b->groups)
    if (!g0.streams.empty() && g0.plan->arch == ARCH_WAVENET && g0.state_family >= 0 && g0.state_family != persist_family(b, g0))
      return false; // (the copying path reports the layout clash)
  return true;
}


// Returns 0 when the buffer went through the session, 1 when the mode does not apply (not enabled / not eligible /
// n_frames not a multiple of 64), < 0 on failure.
int process_host_mapped(nam_hip_batch* b, const float* in_f32, const double* in_f64, float* out_f32, double* out_f64, int n_frames)
{
  if (!host_mapped_applies(b, n_frames) || !host_windows(b, 1, b->in_bar, b->h_out_map, b->d_out_map, b->map_failed))
    return 1;
  const int ic = b->model->spec->in_channels(), oc = b->model->spec->out_channels();
  const long stride = b->max_frames;
  b->pipe_session = false; // (tickets in flight live in windows of their own: this call's window ends their session, flushed)
  const size_t rows_in = (size_t)b->n_streams * ic, rows_out = (size_t)b->n_streams * oc;
  for (size_t r = 0; r < rows_in; r++)
  {
    float* dst = b->in_bar + r * stride;
    if (in_f32)
      copy_to_window(dst, in_f32 + r * n_frames, (size_t)n_frames);
    else // double -> float exactly as _set_condition_array does (NAM/wavenet/model.cpp:817)
      for (int i = 0; i < n_frames; i++)
        dst[i] = (float)in_f64[r * n_frames + i];
  }
  push_out_host_stores();
  b->one_buffer_call = n_frames == kBlock; // (one command, then the caller waits: the stages of a pipeline would only queue up)
  b->short_blocking_call = n_frames <= 4 * kBlock;
  // nam::DSP::process back to back (NAM/dsp.h:97; tools/benchmodel.cpp:129-132: a loop of blocking calls): when the previous
  // call returned a moment ago, the launch this call starts — or still finds — publishes every command's completion and
  // lingers for the next one. A caller that comes once per audio period (1.3 ms at 64 frames) never makes a launch linger.
  const double t_call = stat_now_us();
  const bool linger_now = b->blocking_linger_us > 0 && t_call - b->t_blocking_return < (double)b->blocking_linger_gap_us;
  if (linger_now != b->blocking_linger && b->ps.active && b->ps.outstanding)
  {
    // (the running launch was started under the other rule: let it go first — a whole flush; rare: the pattern changed)
    const int rf = persist_flush(b, b->stream);
    if (rf != NAM_HIP_OK)
      return rf;
  }
  b->blocking_linger = linger_now;
  const int rc = persist_submit(b, b->in_bar, b->d_out_map, n_frames, stride, b->stream);
  if (rc != NAM_HIP_OK)
  {
    b->one_buffer_call = b->short_blocking_call = false;
    return rc < 0 ? rc : fail(NAM_HIP_ERR_DEVICE, "persistent session: the host-mapped buffer was refused");
  }
  // this call's own commands: the per-command completion word when the launch publishes it (it may linger on), else the
  // whole launch (it leaves when it has drained the ring)
  const int rw = b->ps.cmd_done_published ? persist_wait(b, b->stream, b->ps.seq, false) : persist_flush(b, b->stream);
  b->one_buffer_call = b->short_blocking_call = false;
  if (rw != NAM_HIP_OK)
    return rw;
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  for (size_t r = 0; r < rows_out; r++)
  {
    const float* src = b->h_out_map + r * stride;
    if (out_f32)
      std::memcpy(out_f32 + r * n_frames, src, (size_t)n_frames * sizeof(float));
    else
      for (int i = 0; i < n_frames; i++)
        out_f64[r * n_frames + i] = (double)src[i];
  }
  b->t_blocking_return = stat_now_us();
  return NAM_HIP_OK;
}

// ---- ticketed host buffers (include/nam_hip.h: nam_hip_batch_submit_f32 / nam_hip_batch_wait_f32) ----
int pipe_submit(nam_hip_batch* b, const float* in, int n_frames, PipeSlot& sl, int slot)
{
  const int ic = b->model->spec->in_channels(), oc = b->model->spec->out_channels();
  const size_t rows_in = (size_t)b->n_streams * ic, rows_out = (size_t)b->n_streams * oc;
  sl.n_frames = n_frames;
  if (host_mapped_applies(b, n_frames) && host_windows(b, NAM_HIP_PIPE_SLOTS, b->pipe_in_bar, b->pipe_h_out_map, b->pipe_d_out_map, b->pipe_map_failed))
  {
    // the session: the input goes through the PCIe window into the slot's rows, the commands follow it; the resident
    // launch writes the slot's rows of the host-side window
    const long stride = b->max_frames, at = (long)slot * (long)std::max(rows_in, rows_out) * b->max_frames;
    const double t0 = stats_on() ? stat_now_us() : 0.0;
    for (size_t r = 0; r < rows_in; r++)
      copy_to_window(b->pipe_in_bar + at + r * stride, in + r * n_frames, (size_t)n_frames);
    push_out_host_stores();
    const double t1 = stats_on() ? stat_now_us() : 0.0;
    b->pipe_session = true;
    const int rc = persist_submit(b, b->pipe_in_bar + at, b->pipe_d_out_map + at, n_frames, stride, b->stream);
    if (stats_on())
    {
      const double t2 = stat_now_us();
      b->ps.t_in += t1 - t0, b->ps.t_cmd += t2 - t1;
      b->ps.t_in_max = std::max(b->ps.t_in_max, t1 - t0), b->ps.t_cmd_max = std::max(b->ps.t_cmd_max, t2 - t1);
    }
    if (rc != NAM_HIP_OK)
      return rc < 0 ? rc : fail(NAM_HIP_ERR_DEVICE, "persistent session: the host-mapped buffer was refused");
    sl.how = 0;
    sl.seq_end = b->ps.seq;
    sl.epoch = b->ps.epoch;
    return NAM_HIP_OK;
  }
  if (!b->ps.enabled || !persist_eligible(b))
  {
    // launches on the batch's stream: pinned staging in, copy, launch, copy, pinned staging out — all enqueued, an event behind them
    const size_t slot_in = rows_in * b->max_frames, slot_out = rows_out * b->max_frames;
    if (!b->pipe_h_in)
    {
      NAM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&b->pipe_h_in), slot_in * NAM_HIP_PIPE_SLOTS * sizeof(float), hipHostMallocDefault));
      NAM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&b->pipe_h_out), slot_out * NAM_HIP_PIPE_SLOTS * sizeof(float), hipHostMallocDefault));
      NAM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&b->pipe_d_in), slot_in * NAM_HIP_PIPE_SLOTS * sizeof(float)));
      NAM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&b->pipe_d_out), slot_out * NAM_HIP_PIPE_SLOTS * sizeof(float)));
    }
    if (!sl.done)
      NAM_HIP_CHECK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    float *hi = b->pipe_h_in + slot * slot_in, *ho = b->pipe_h_out + slot * slot_out;
    float *di = b->pipe_d_in + slot * slot_in, *dn = b->pipe_d_out + slot * slot_out;
    std::memcpy(hi, in, rows_in * n_frames * sizeof(float));
    NAM_HIP_CHECK(hipMemcpyAsync(di, hi, rows_in * n_frames * sizeof(float), hipMemcpyHostToDevice, b->stream));
    const int rc = nam_hip_batch_process_device(b, di, dn, n_frames, n_frames, nullptr);
    if (rc != NAM_HIP_OK)
      return rc;
    NAM_HIP_CHECK(hipMemcpyAsync(ho, dn, rows_out * n_frames * sizeof(float), hipMemcpyDeviceToHost, b->stream));
    NAM_HIP_CHECK(hipEventRecord(sl.done, b->stream));
    sl.how = 1;
    return NAM_HIP_OK;
  }
  // a session batch with a ragged length (or without host-mapped memory): rendered now, handed out by the wait
  sl.held.resize(rows_out * n_frames);
  const int rc = nam_hip_batch_process_f32(b, in, sl.held.data(), n_frames);
  if (rc != NAM_HIP_OK)
    return rc;
  sl.how = 2;
  return NAM_HIP_OK;
}

int pipe_wait(nam_hip_batch* b, PipeSlot& sl, int slot, float* out)
{
  const int oc = b->model->spec->out_channels();
  const size_t rows_out = (size_t)b->n_streams * oc;
  const int n_frames = sl.n_frames;
  if (sl.how == 0)
  {
    const double t0 = stats_on() ? stat_now_us() : 0.0;
    if (b->ps.active && sl.epoch == b->ps.epoch) // (a session that has ended ended flushed)
    {
      const int rc = persist_wait(b, b->stream, sl.seq_end, false);
      if (rc != NAM_HIP_OK)
        return rc;
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    const double t1 = stats_on() ? stat_now_us() : 0.0;
    if (out)
    {
      const size_t rows_in = (size_t)b->n_streams * b->model->spec->in_channels();
      const long stride = b->max_frames, at = (long)slot * (long)std::max(rows_in, rows_out) * b->max_frames;
      for (size_t r = 0; r < rows_out; r++)
        std::memcpy(out + r * n_frames, b->pipe_h_out_map + at + r * stride, (size_t)n_frames * sizeof(float));
    }
    if (stats_on())
    {
      const double t2 = stat_now_us();
      b->ps.t_poll += t1 - t0, b->ps.t_out += t2 - t1;
      b->ps.t_poll_max = std::max(b->ps.t_poll_max, t1 - t0), b->ps.t_out_max = std::max(b->ps.t_out_max, t2 - t1);
    }
  }
  else if (sl.how == 1)
  {
    NAM_HIP_CHECK(hipEventSynchronize(sl.done));
    if (out)
      std::memcpy(out, b->pipe_h_out + (size_t)slot * rows_out * b->max_frames, rows_out * n_frames * sizeof(float));
  }
  else if (out)
    std::memcpy(out, sl.held.data(), rows_out * n_frames * sizeof(float));
  sl.in_flight = false;
  return NAM_HIP_OK;
}

} // namespace api
} // namespace namhip
