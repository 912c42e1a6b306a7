// wr_jit.h — nam_wn_reg_kernel compiled for one model's own layer shapes, at load time, with an on-disk cache.
#pragma once

#include <string>

#include "plan.h"

namespace namhip
{

// Compiles kernel_wn_reg.hip with `shapes` as its shape tables (WrShapeSet::header_text) into a gfx950 code object —
// or finds the one a previous load of the same shapes (by any process) left in the cache. Host-only: needs hipcc and the
// kernel sources, not a GPU. Returns the code object's path; "" on failure with the reason in `why`.
//   sources:  $NAM_HIP_JIT_SRC, else <directory of libnam_hip.so>/../csrc (the in-tree layout)
//   compiler: $NAM_HIP_HIPCC, else /opt/rocm/bin/hipcc
//   cache:    $NAM_HIP_JIT_CACHE, else <directory of libnam_hip.so>/jit (in-tree: travels with the library), else
//             /tmp/nam_hip_jit_<uid>
//   NAM_HIP_JIT=0 switches the per-model compile off (models outside the ahead-of-time shapes then run the
//   run-time-flag instantiations or the op interpreter).
std::string wr_jit_build(const WrShapeSet& shapes, std::string& why);
bool wr_jit_enabled();

} // namespace namhip
