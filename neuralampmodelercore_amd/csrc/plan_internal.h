// plan_internal.h — what the planner's translation units (plan.cpp, plan_ops.cpp, plan_a1.cpp, plan_wr.cpp) call across each other.
#pragma once
#include "plan.h"
#include "kp_table.h"
#include "aq_table.h"

namespace namhip
{
// plan_ops.cpp
int build_op_program(const WaveNetSpec& wn, Plan& plan);
// plan_a1.cpp
bool a1_channel_supported(int c);
void build_a1_ws(const WaveNetSpec& wn, Plan& plan);
void build_a1_il(Plan& plan);
void build_a1(const WaveNetSpec& wn, Plan& plan);
void build_a1_kt(Plan& plan);
void build_a1_kp(Plan& plan);
void build_a1_q(Plan& plan);
bool official_standard_topology(const WaveNetSpec& wn);
bool pad_channels_for_mfma(const WaveNetSpec& wn, WaveNetSpec& out);
void validate_wavenet_geometry(const WaveNetSpec& wn);
// plan_wr.cpp
void build_wr(const WaveNetSpec& wn, Plan& plan, WrShapeSet* jit_shapes);
} // namespace namhip
