// NAM/slimmable.h — C++ adapter: the reference's `nam::SlimmableModel` interface (reference NAM/slimmable.h:13-29), a header of
// its own because callers name it (tools/benchmodel.cpp:11, tools/render.cpp:14). The implementations are the adapter's
// DSP classes (NAM/dsp.h), which forward to nam_hip_batch_set_slimmable_size.
#pragma once

#include <vector>

namespace nam
{

class SlimmableModel
{
public:
  virtual ~SlimmableModel() = default;
  // 0.0 (smallest sub-model) .. 1.0 (the full model); not for the audio thread
  virtual void SetSlimmableSize(const double val) = 0;
  // sorted breakpoints in (0, 1) between the selectable sub-models; 0 and 1 are implied
  virtual std::vector<double> GetSlimmableSizeBreakpoints() const { return {}; }
};

} // namespace nam
