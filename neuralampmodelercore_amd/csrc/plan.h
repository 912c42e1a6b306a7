// plan.h — the device "plan": what the HIP kernels execute for one model.
//
// A plan is produced once per model on the host (plan.cpp) from a ModelSpec:
//   * `ops`     a flat micro-op program, interpreted by the generic WaveNet kernel. One wavefront
//               runs the whole program for one (stream, 64-frame block); lane = frame. Activations
//               live in LDS as rows of 64 floats ("row" = one channel x 64 frames).
//   * `blob`    every weight the ops need, re-laid-out for the kernels (dense, padded, tap-major),
//               from the reference's flat weight stream (order: NAM/wavenet/model.cpp:152-181,
//               563-569, 661-683; conv layouts NAM/conv1d.cpp:40-55, NAM/dsp.cpp:384-397).
//   * state     per-stream persistent state layout in HBM: for every dilated conv a history ring
//               `[cin][R]` (time contiguous => a tap read is 64 consecutive floats) plus its write
//               position. This replaces nam::RingBuffer (NAM/ring_buffer.cpp:7-109).
//   * `a1`      (optional) description for the specialised register-resident kernel used for the
//               plain "A1" WaveNet family (ungated, no FiLM, groups=1): wavenet_a1_standard.nam.
//   * `lstm`    (optional) description for the LSTM kernel (NAM/lstm.cpp:31-168).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "model_spec.h"

namespace namhip
{

constexpr int kBlock = 64; // frames per device block == wavefront width

enum OpType : int32_t
{
  OP_END = 0,
  OP_LOAD_IN = 1, // dst rows <- input channels
  OP_STORE_OUT = 2, // output channels <- src rows (* blob[w] if w >= 0)
  OP_CONV = 3, // dst = [bias +] sum_k sum_ci W[k][ci][co] * tap_k(src)[ci]   (k = 1: pointwise)
  OP_FILM = 4, // dst[c] = src[c] * aux[c] (+ aux[cout + c] if flag)  — aux = scale/shift rows
  OP_ACT = 5, // in-place activation on dst, cout channels
  OP_GATE = 6, // gated / blended activation on a 2B-row buffer -> top B rows
  OP_ADD = 7, // dst = src + aux
  OP_COPY = 8, // dst = src
  OP_ZERO = 9, // dst = 0
  OP_SCALE = 10 // dst = blob[w] * src
};

struct NamOp // 16 x int32 = 64 bytes, fetched with one scalar load
{
  int32_t type;
  int32_t dst; // LDS float offset of row 0 of the destination
  int32_t src; // LDS float offset of the source
  int32_t aux; // LDS float offset of the second operand
  int32_t cin;
  int32_t cout;
  int32_t cout_pad; // conv: padded output count (multiple of cb)
  int32_t cb; // conv: output-channel register block (4 or 8)
  int32_t w; // blob offset (floats): conv weights [k][cin][cout_pad] / act params / scale
  int32_t b; // blob offset of bias (conv), second act params (gate), -1 = none
  int32_t k; // conv: kernel size; act: activation type
  int32_t dil; // conv: dilation; gate: secondary activation type
  int32_t state; // conv: float offset of this conv's ring inside the per-stream state, -1 = no ring
  int32_t ring; // conv: ring length R (frames); act/gate: number of PReLU slopes (primary)
  int32_t ring_id; // conv: index into the per-stream write-position table; gate: #slopes (secondary)
  int32_t flag; // FILM: 1 = shift present; GATE: 1 = gated, 2 = blended
};
static_assert(sizeof(NamOp) == 64, "NamOp must stay 64 bytes");

// ---- A1-family fast path -------------------------------------------------------------------
// Per layer-array description for the specialised kernel; weights live in `blob` at `w_base`:
//   rechannel  [in_size][C]                           (no bias)
//   per layer: conv W [K][C][C] (tap-major, ci, co), conv bias [C], mixin [C], W1x1 [C][C] (ci, co),
//              b1x1 [C]
//   head rechannel [C][H] (ci, co), head bias [H] (zeros if absent)
constexpr int kA1MaxArrays = 4;
constexpr int kA1MaxLayers = 32;

struct A1Array
{
  int32_t in_size; // rechannel input channels (1 for array 0, previous C otherwise)
  int32_t channels; // C
  int32_t kernel; // K (uniform inside the array)
  int32_t n_layers;
  int32_t head_size; // H
  int32_t act; // ActType
  int32_t w_base; // blob offset of this array's packed weights
  int32_t layer_stride; // floats per layer in the packed weights
  int32_t state_base; // float offset of layer 0's ring inside the per-stream state
  int32_t dil[kA1MaxLayers];
  int32_t ring_off[kA1MaxLayers]; // float offset (from stream state base) of each layer's ring
  int32_t ring_len[kA1MaxLayers]; // R = (K-1)*d + 64
  int32_t ring_id[kA1MaxLayers];
};

struct A1Plan
{
  int32_t valid = 0;
  int32_t n_arrays = 0;
  int32_t head_scale_off = 0; // blob offset
  int32_t n_rings = 0;
  int32_t ring_len_by_id[64]; // R of ring r (for the per-block write-position update)
  A1Array arr[kA1MaxArrays];
};

// ---- LSTM ------------------------------------------------------------------------------------
// blob layout per layer: W [4H][I+H] row-major, b [4H]; then head W [out][H], head b [out].
// Per-stream state: for each layer h[H], c[H] (initialised from the weight stream, lstm.cpp:24-28).
struct LSTMPlan
{
  int32_t valid = 0;
  int32_t n_layers = 0, input_size = 0, hidden = 0, in_ch = 1, out_ch = 1;
  int32_t fast = 0; // activations::Activation::using_fast_tanh (lstm.cpp:48)
  int32_t head_w = 0, head_b = 0;
  int32_t layer_w[16] = {0};
  int32_t layer_b[16] = {0};
  std::vector<float> init_state; // [n_layers][2][H]
};

struct Plan
{
  int arch = 0;
  int in_channels = 1, out_channels = 1;
  int prewarm_samples = 0;
  std::vector<NamOp> ops;
  std::vector<float> blob;
  int lds_rows = 0; // LDS rows (x 64 floats) the generic kernel needs per wavefront
  int n_rings = 0;
  int state_floats = 0; // per-stream state size (floats), multiple of 64; first n_rings words = write positions
  A1Plan a1;
  LSTMPlan lstm;
  std::string describe() const;
};

// Build the plan for a (non-slimmable view of a) model. Throws std::runtime_error on unsupported shapes.
Plan build_plan(const ModelSpec& model);
Plan build_wavenet_plan(const WaveNetSpec& wn);

} // namespace namhip
