// NAM/dsp.h — C++ adapter: the reference's `nam::DSP` surface on top of the nam_hip C ABI.
//
// Source compatibility shim for callers of sdatkinson/NeuralAmpModelerCore (the plugin,
// tools/benchmodel.cpp, tools/render.cpp): same class name, same virtual signatures
// (reference NAM/dsp.h:70-231), but `process` forwards to HIP kernels on an MI355X through
// libnam_hip.so (include/nam_hip.h). One `nam::DSP` object = a batch of ONE stream; for the
// many-stream hot path use `nam::BatchDSP` below (or the C ABI directly).
#pragma once

#include <atomic>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/nam_hip.h"
#include "slimmable.h"

#ifdef NAM_SAMPLE_FLOAT
  #define NAM_SAMPLE float
#else
  #define NAM_SAMPLE double
#endif
#ifndef NAM_DEFAULT_MAX_BUFFER_SIZE
  #define NAM_DEFAULT_MAX_BUFFER_SIZE 4096
#endif
#define NAM_UNKNOWN_EXPECTED_SAMPLE_RATE -1.0

namespace nam
{

// nam::NamFileValidationError (reference NAM/nam_file.h:11)
class NamFileValidationError : public std::runtime_error
{
public:
  using std::runtime_error::runtime_error;
};

namespace activations
{
// The reference switches fast tanh through a process-global (NAM/activations.cpp:168-187); here the
// flag is read once, when a model is loaded, and passed to nam_hip_model_load.
class Activation
{
public:
  static void enable_fast_tanh() { using_fast_tanh = true; }
  static void disable_fast_tanh() { using_fast_tanh = false; }
  static inline bool using_fast_tanh = false;
};
} // namespace activations

namespace detail
{
inline void check(int rc)
{
  if (rc >= 0)
    return;
  const std::string msg = nam_hip_last_error();
  if (rc == NAM_HIP_ERR_FILE)
    throw NamFileValidationError(msg);
  throw std::runtime_error(msg);
}
struct ModelDeleter
{
  void operator()(nam_hip_model* m) const { nam_hip_model_free(m); }
};
struct BatchDeleter
{
  void operator()(nam_hip_batch* b) const { nam_hip_batch_destroy(b); }
};
} // namespace detail

namespace detail
{
// the thread-local default newly constructed DSP objects copy (reference NAM/dsp.cpp:20)
inline bool& prewarm_on_reset_default()
{
  static thread_local bool value = true;
  return value;
}
} // namespace detail

// Scoped change of that default (reference NAM/dsp.h:40-56): objects constructed on this thread while it lives take the
// scoped value as their instance setting; existing objects are untouched.
class ScopedPrewarmOnResetDefault
{
public:
  explicit ScopedPrewarmOnResetDefault(const bool prewarmOnReset)
  : mPreviousPrewarmOnReset(detail::prewarm_on_reset_default())
  {
    detail::prewarm_on_reset_default() = prewarmOnReset;
  }
  ~ScopedPrewarmOnResetDefault() { detail::prewarm_on_reset_default() = mPreviousPrewarmOnReset; }
  ScopedPrewarmOnResetDefault(const ScopedPrewarmOnResetDefault&) = delete;
  ScopedPrewarmOnResetDefault& operator=(const ScopedPrewarmOnResetDefault&) = delete;
  bool PreviousPrewarmOnReset() const { return mPreviousPrewarmOnReset; }

private:
  bool mPreviousPrewarmOnReset;
};

class DSP
{
public:
  DSP(std::shared_ptr<nam_hip_model> model, int device = 0)
  : mModel(std::move(model))
  , mDevice(device)
  , mPrewarmOnReset(detail::prewarm_on_reset_default())
  {
    detail::check(nam_hip_model_get_info(mModel.get(), &mInfo));
    mInputLevel = mInfo.input_level;
    mOutputLevel = mInfo.output_level;
    mLoudness = mInfo.loudness;
    mHasInputLevel = mInfo.has_input_level != 0;
    mHasOutputLevel = mInfo.has_output_level != 0;
    mHasLoudness = mInfo.has_loudness != 0;
  }
  virtual ~DSP() = default;

  virtual void prewarm()
  {
    if (mMaxBufferSize == 0)
      SetMaxBufferSize(NAM_DEFAULT_MAX_BUFFER_SIZE);
    detail::check(nam_hip_batch_reset(mBatch.get(), 1));
  }

  // input[channel][frame], output[channel][frame] — reference NAM/dsp.h:97
  virtual void process(NAM_SAMPLE** input, NAM_SAMPLE** output, const int num_frames)
  {
    if (!mBatch)
      SetMaxBufferSize(num_frames > NAM_DEFAULT_MAX_BUFFER_SIZE ? num_frames : NAM_DEFAULT_MAX_BUFFER_SIZE);
    const int ic = NumInputChannels(), oc = NumOutputChannels();
    if (num_frames > mMaxBufferSize) // (the staging buffers are sized in SetMaxBufferSize: no allocation on the audio path)
      throw std::runtime_error("process: num_frames exceeds the max buffer size set by Reset()");
    for (int c = 0; c < ic; c++)
      for (int i = 0; i < num_frames; i++)
        mIn[(size_t)c * num_frames + i] = input[c][i];
#ifdef NAM_SAMPLE_FLOAT
    detail::check(nam_hip_batch_process_f32(mBatch.get(), mIn.data(), mOut.data(), num_frames));
#else
    detail::check(nam_hip_batch_process_f64(mBatch.get(), mIn.data(), mOut.data(), num_frames));
#endif
    for (int c = 0; c < oc; c++)
      for (int i = 0; i < num_frames; i++)
        output[c][i] = mOut[(size_t)c * num_frames + i];
  }

  double GetExpectedSampleRate() const { return mInfo.expected_sample_rate; }
  int NumInputChannels() const { return mInfo.in_channels; }
  int NumOutputChannels() const { return mInfo.out_channels; }
  double GetInputLevel() { return mInputLevel; }
  double GetLoudness() const
  {
    if (!HasLoudness())
      throw std::runtime_error("Asked for loudness of a model that doesn't know how loud it is!");
    return mLoudness;
  }
  double GetOutputLevel() { return mOutputLevel; }
  bool HasInputLevel() { return mHasInputLevel; }
  bool HasLoudness() const { return mHasLoudness; }
  bool HasOutputLevel() { return mHasOutputLevel; }
  virtual int GetPrewarmSamples() { return mInfo.prewarm_samples; }

  virtual void Reset(const double sampleRate, const int maxBufferSize)
  {
    mExternalSampleRate = sampleRate;
    mHaveExternalSampleRate = true;
    SetMaxBufferSize(maxBufferSize);
    detail::check(nam_hip_batch_reset(mBatch.get(), GetPrewarmOnReset() ? 1 : 0));
  }
  void ResetAndPrewarm(const double sampleRate, const int maxBufferSize)
  {
    const bool prev = GetPrewarmOnReset();
    SetPrewarmOnReset(true);
    Reset(sampleRate, maxBufferSize);
    SetPrewarmOnReset(prev);
  }
  virtual void SetPrewarmOnReset(const bool prewarmOnReset) { mPrewarmOnReset = prewarmOnReset; }
  bool GetPrewarmOnReset() const { return mPrewarmOnReset; }
  void SetInputLevel(const double v)
  {
    mInputLevel = v;
    mHasInputLevel = true;
  }
  void SetLoudness(const double v)
  {
    mLoudness = v;
    mHasLoudness = true;
  }
  void SetOutputLevel(const double v)
  {
    mOutputLevel = v;
    mHasOutputLevel = true;
  }
  int GetMaxBufferSize() const { return mMaxBufferSize; }
  nam_hip_batch* GetBatchHandle() { return mBatch.get(); }

protected:
  virtual void SetMaxBufferSize(const int maxBufferSize)
  {
    if (mBatch && maxBufferSize == mMaxBufferSize)
      return;
    nam_hip_batch* b = nullptr;
    detail::check(nam_hip_batch_create(mModel.get(), mDevice, NumStreams(), maxBufferSize, &b));
    mBatch.reset(b);
    // persistent block mode where the model's kernel has one (include/nam_hip.h): process() then costs a command and a
    // wait on host-mapped buffers instead of two copies, a launch and a stream synchronise per buffer (buffers of a
    // multiple of 64 frames; anything else falls back to a launch, transparently)
    (void)nam_hip_batch_set_persistent(b, 1);
    mMaxBufferSize = maxBufferSize;
    mIn.assign((size_t)NumInputChannels() * NumStreams() * maxBufferSize, NAM_SAMPLE(0));
    mOut.assign((size_t)NumOutputChannels() * NumStreams() * maxBufferSize, NAM_SAMPLE(0));
  }
  virtual int NumStreams() const { return 1; }

  std::shared_ptr<nam_hip_model> mModel;
  std::unique_ptr<nam_hip_batch, detail::BatchDeleter> mBatch;
  nam_hip_model_info mInfo{};
  int mDevice = 0;
  int mMaxBufferSize = 0;
  bool mHaveExternalSampleRate = false;
  double mExternalSampleRate = -1.0;
  std::atomic<bool> mPrewarmOnReset;
  bool mHasLoudness = false, mHasInputLevel = false, mHasOutputLevel = false;
  double mLoudness = 0.0, mInputLevel = 0.0, mOutputLevel = 0.0;
  std::vector<NAM_SAMPLE> mIn, mOut;
};

// A slimmable .nam: dynamic_cast<nam::SlimmableModel*>(dsp.get()) succeeds, as in the reference
// (tools/benchmodel.cpp:93, tools/render.cpp:119).
class SlimmableDSP : public DSP, public SlimmableModel
{
public:
  using DSP::DSP;
  void SetSlimmableSize(const double val) override
  {
    mRatio = val;
    if (mBatch)
      detail::check(nam_hip_batch_set_slimmable_size(mBatch.get(), nullptr, 0, val));
  }
  std::vector<double> GetSlimmableSizeBreakpoints() const override
  {
    double buf[64];
    const int n = nam_hip_model_slimmable_breakpoints(mModel.get(), buf, 64);
    detail::check(n);
    return std::vector<double>(buf, buf + (n < 64 ? n : 64));
  }

protected:
  void SetMaxBufferSize(const int maxBufferSize) override
  {
    DSP::SetMaxBufferSize(maxBufferSize);
    detail::check(nam_hip_batch_set_slimmable_size(mBatch.get(), nullptr, 0, mRatio));
  }
  double mRatio = 1.0;
};

// The many-stream form of the same object (not in the reference): N independent streams, planar
// float32 host buffers [stream][channel][frame].
class BatchDSP : public DSP
{
public:
  BatchDSP(std::shared_ptr<nam_hip_model> model, int n_streams, int device = 0)
  : DSP(std::move(model), device)
  , mStreams(n_streams)
  {
  }
  void process_batch(const float* in, float* out, const int num_frames)
  {
    detail::check(nam_hip_batch_process_f32(mBatch.get(), in, out, num_frames));
  }
  // process_batch in two halves, for callers that keep buffers in flight (up to NAM_HIP_PIPE_SLOTS): submit copies `in`
  // and returns a ticket at once, wait blocks until that buffer's output is in `out` (include/nam_hip.h)
  int64_t submit(const float* in, const int num_frames)
  {
    int64_t ticket = -1;
    detail::check(nam_hip_batch_submit_f32(mBatch.get(), in, num_frames, &ticket));
    return ticket;
  }
  void wait(const int64_t ticket, float* out) { detail::check(nam_hip_batch_wait_f32(mBatch.get(), ticket, out)); }
  void SetSlimmableSize(const int* stream_ids, int n, double ratio)
  {
    detail::check(nam_hip_batch_set_slimmable_size(mBatch.get(), stream_ids, n, ratio));
  }
  bool IsSlimmable() const { return mInfo.is_slimmable != 0; }
  // Device-resident audio (a server that keeps its streams' buffers in HBM): planar float32 DEVICE pointers
  // [stream][channel][frame_stride]; enqueue-only. In the persistent block mode (on since Reset) a call of a multiple of 64
  // frames is num_frames / 64 commands of the session and costs the host ~1.5 us; flush() returns when every submitted
  // buffer is rendered and visible. `hip_stream`: the hipStream_t the input was produced on (nullptr = the batch's own).
  void process_device(const float* d_in, float* d_out, const int num_frames, const int64_t frame_stride, void* hip_stream = nullptr)
  {
    detail::check(nam_hip_batch_process_device(mBatch.get(), d_in, d_out, num_frames, frame_stride, hip_stream));
  }
  void flush(void* hip_stream = nullptr) { detail::check(nam_hip_batch_flush(mBatch.get(), hip_stream)); }
  void synchronize() { detail::check(nam_hip_batch_synchronize(mBatch.get())); }
  // Offline re-amp of one whole signal per stream (lengths may differ); planar float32 host buffers.
  void render(const float* const* in, float* const* out, const int64_t* n_frames)
  {
    detail::check(nam_hip_batch_render_f32(mBatch.get(), in, out, n_frames));
  }

protected:
  int NumStreams() const override { return mStreams; }
  int mStreams;
};

} // namespace nam
