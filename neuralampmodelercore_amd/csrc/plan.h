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
//               `[R][cin]` (frame-major: one frame's channels are contiguous, so a lane that owns a
//               frame moves its channels with 16-byte accesses) plus its write position.
//               This replaces nam::RingBuffer (NAM/ring_buffer.cpp:7-109).
//   * `a1`      (optional) description for the specialised register-resident kernel used for the
//               plain "A1" WaveNet family (ungated, no FiLM, groups=1): wavenet_a1_standard.nam.
//   * `lstm`    (optional) description for the LSTM kernel (NAM/lstm.cpp:31-168).
#pragma once

#include <cstdint>
#include <array>
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
  OP_SCALE = 10, // dst = blob[w] * src
  OP_STAGE = 11 // block start: the last `k` frames (= lookback) of ring `state` -> LDS floats [dst, dst + cin * k), frame-major;
                // emitted as a run (cout of the first = length of the run) so that every ring's rows are requested before
                // the first one is waited for: one memory round trip per block instead of one per tap and channel
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
  int32_t flag; // conv with FiLM epilogue: 1 = scale, 2 = scale + shift; GATE: 1 = gated, 2 = blended;
                // conv with a ring: 4 = history staged in LDS by OP_STAGE at float offset `aux` (frame-major [lookback][cin])
};
static_assert(sizeof(NamOp) == 64, "NamOp must stay 64 bytes");

// ---- A1-family fast path -------------------------------------------------------------------
// Per layer-array description for the specialised kernel; weights live in `blob` at `w_base`:
//   rechannel  [in_size][C]                           (no bias)
//   per layer: conv W [K_l][C][C] (tap-major, ci, co), conv bias [C], mixin [C], W1x1 [C][C] (ci, co),
//              b1x1 [C]
//   head rechannel [K_h][C][H] (tap-major, ci, co), head bias [H] (zeros if absent)
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
  // per-layer kernel sizes (A2 mixes K = 6 and 15) and where each layer's packed weights start (floats from w_base)
  int32_t ksize[kA1MaxLayers];
  int32_t layer_off[kA1MaxLayers];
  // head rechannel = Conv1D(K = head_k, dilation head_dil) over the head accumulator: packed [k][c][h] at head_off
  // (floats from w_base), bias [H] behind it; with head_k > 1 it has its own history ring (A2: K = 16)
  int32_t head_k, head_dil, head_off;
  int32_t head_ring_off, head_ring_len, head_ring_id; // id -1: no ring
  float act_p0; // first activation parameter (LeakyReLU slope)
  int32_t pad0;
};

// LDS geometry of nam_a1_mfma_kernel's history buffers (shared with the host, which precomputes every
// per-job LDS offset)
constexpr int kMfSC = 24; // floats per frame row: 6 sixteen-byte slots => conflict-free ds_read_b128 for lane (g, j)
constexpr int kMfXwFloats = 2 * kBlock * kMfSC; // one window: [previous 64 | current 64] frames
constexpr int kMfTbFloats = kBlock * kMfSC; // one tap buffer: 64 frames
constexpr int kMfXwOff = 0; // window [2 buffers]
constexpr int kMfTbOff = kMfXwOff + 2 * kMfXwFloats; // tap buffers [2 buffers][2 taps]

// ---- MFMA kernel (nam_a1_mfma_kernel, wave-specialised): one job per LAYER -------------------------------
// The rechannel / head-rechannel steps ride on the neighbouring layer jobs (one extra weight tile + one
// extra constant vector per job), so a block is exactly n_layers jobs (+1 idle job when that is odd: the
// LDS double buffers alternate per job and must come back to parity 0 at the start of every block).
// Compute waves (0-3) and mover waves (4-7) execute from separate descriptors.
// Mover prefetch depth D (jobs of history loads in flight = the mover loop's unroll factor) is chosen per model
// so that it divides the jobs per block: then a launch is an exact number of unrolled bodies and no idle jobs
// pad its tail (one 64-frame launch of a 20-job model would otherwise run 24). Instantiated: 5 and 6.
constexpr int kWsPrefetchMax = 6;
constexpr int kWsJobMax = 32; // jobs (layers) per block
constexpr int kWsXtMax = 8; // extra tiles (rechannel / head rechannel matrices) per model
constexpr int kWsTileFloats = 4 * 256; // per job: [conv tap 0,1,2 | layer1x1][lane][4 k-steps]
// LDS (floats): window [2][128][SC] | taps [2][2][64][SC] | consts [jobs][64] | tiles [2][1024] |
//               extra tiles [n][256] | input samples [2][64]        (dynamic: sized per model, see ws_lds_*)
constexpr int kWsConstsOff = kMfTbOff + 4 * kMfTbFloats;

enum CDescFlags : int32_t
{
  CD_LAYER = 1,
  CD_X0 = 2, // first job of a block: x = rechannel column (extra consts) * input sample, head = 0
  CD_PRE_HEAD = 4, // first layer of a later array: head = HeadW(prev array) . head + head bias (extra tile / consts)
  CD_POST_RECH = 8, // last layer of a non-final array: x = RechW(next array) . x (extra tile) before publishing
  CD_POST_OUT = 16, // last layer of the final array: out = head_scale * (HeadW . head + head bias)[0]
  CD_HALF = 32, // 8-channel array: duplicated / rotated lane layout, two k-steps per matrix (plan.cpp)
  CD_PREV_HALF = 64 // CD_PRE_HEAD: the previous array's head accumulator is in the half layout
};
struct CDesc // 8 x int32, one s_load_dwordx8
{
  int32_t flags;
  int32_t act;
  int32_t gp; // bits 0-7: 16 * (C/4 - 1), clamp for the lane's channel-quad byte offset when reading operands;
              // bits 8-15: 16 * (C_published/4 - 1), lanes beyond do not publish
  int32_t consts_b; // LDS byte offset of this job's 64 constants: bias | mixin | 1x1 bias | extra
  int32_t tap0_b, tap1_b; // LDS byte offset of frame 0 of tap k's operand rows
  int32_t pub_b; // LDS byte offset of frame 0 of the window rows x is published to
  int32_t xt_b; // LDS byte offset of this job's extra tile (any valid tile when it has none)
};
enum VDescFlags : int32_t
{
  MV_RING = 1, // append this job's input (window rows of its buffer) to its history ring
  MV_SUCC_FIRST = 2, // the successor is the next block's first job: also drop x0 = rechannel * input and the input itself
  MV_SUCC_B = 4 // the successor has a second history set
};
// A job reads at most two 64-frame history sets from its ring (lookbacks LA >= LB, in frames):
//   2d <= 64      : A = previous block (L = 64) -> window;                  both taps read the window
//   d <= 64 < 2d  : A = previous block -> window, B = tap 0 (L = 2d) -> tap buffer 0
//   d > 64        : A = tap 0 (L = 2d) -> tap buffer 0, B = tap 1 (L = d) -> tap buffer 1
struct VDesc // 16 x int32, one s_load_dwordx16
{
  int32_t flags;
  int32_t st_a_b, st_b_b, st_x0_b; // LDS byte offsets where the SUCCESSOR's sets / x0 rows are dropped
  int32_t f_rbase, f_R, f_LA, f_LB, f_ring_id, f_q16max; // ring geometry of the job prefetched now (ws_prefetch + 1
                                                         // ahead); f_LB == 0: no second set (the load is redirected)
  int32_t ap_src_b; // LDS byte offset of frame 0 of this job's input rows (current half of its window)
  int32_t ring_b, R, ring_id, q16max; // this job's ring
  int32_t pad0;
};
static_assert(sizeof(CDesc) == 32 && sizeof(VDesc) == 64, "descriptor sizes are part of the kernel ABI");

// ---- Interleaved-frame MFMA kernel (nam_a1_p2_kernel and its pipelines): same models, tiles and constants as nam_a1_mfma_kernel ----
// Compute wave w of a stream's workgroup owns frames t = 4 j + w (j = lane & 15) of the 64-frame block instead of 16
// consecutive ones. A dilated tap (t - L) then lives
//   * L >= 64            : in an earlier block — the lane fetches it from the history ring itself (IL_HIST);
//   * L = d or 2d, d in {4, 8, 16, 32}: in the SAME wave, L / 4 lanes down the 16-lane row — a DPP row shift of the
//                          lane's own registers; the lanes that fall off the row take the previous block's frames from
//                          the ring (IL_DPP);
//   * anything else (d = 1, 2, ...): in another wave — one LDS window + one workgroup barrier (IL_EXCH).
// wavenet_a1_standard: 4 barrier jobs per block instead of 20; the other 16 layers run without touching LDS for
// activations and without waiting for anybody. No mover waves: every compute lane requests its own history kIlDepth
// jobs ahead (two 16-byte buffer loads per job, lanes that need nothing use an out-of-range offset = no traffic) and
// appends its own frame to the ring. A fifth "loader" wave copies the constants and the weight tiles of all jobs
// into LDS once per launch, job by job, and publishes its progress in an LDS word the compute waves check.
enum IlKind : int32_t
{
  IL_IDLE = 0, // padding job (jobs per block are padded to a multiple of the request depth)
  IL_HIST = 1,
  IL_DPP = 2,
  IL_EXCH = 3
};
struct IlDesc // 16 x int32, one s_load_dwordx16
{
  int32_t flags; // CDescFlags
  int32_t kind; // IlKind
  int32_t act;
  int32_t gp; // as CDesc::gp, bits 0-7 only: 16 * (C / 4 - 1)
  int32_t ring_b, R, ring_id, row_b; // this job's ring: byte offset in the stream state, frames, position index, C * 4
  int32_t dil;
  int32_t tap0_lds; // IL_EXCH: 1 = tap 0 (lookback 2d <= 64) is read from the LDS window, 0 = from the ring request
  int32_t n_consts_b, n_xt_b, n_tiles_b; // LDS byte offsets of the NEXT job's constants / extra tile / 4 tiles
  int32_t n_ready; // loader progress required before the next job's operands may be read (its job index + 1)
  int32_t pad[2];
};
struct IlFetch // 8 x int32: the two ring requests issued now for the job `depth` ahead
{
  int32_t ring_b, R, ring_id, row_b;
  int32_t LA, LB; // lookbacks in frames (0 = no request)
  int32_t nA, nB; // lanes j < n request (16 = every lane)
};
static_assert(sizeof(IlDesc) == 64 && sizeof(IlFetch) == 32, "descriptor sizes are part of the kernel ABI");
constexpr int kIlJobMax = kWsJobMax + 8;
constexpr int kIlWinRowB = kMfSC * 4; // LDS window row pitch (bytes)
constexpr int kIlWinB = 2 * kBlock * kIlWinRowB; // one window: [previous 64 | current 64] frames

// ---- The official A1 topology at compile time (nam_a1_p2_kernel) ----------------------------------------------
// Every official WaveNet size (standard 16/8, lite 12/6, feather 8/4 channels) is two arrays of ten layers, kernel
// size 3, dilations 1, 2, 4, ... 512. For that topology the whole job table of the interleaved-frame kernel is a
// function of the two (padded) channel counts: these constexpr functions restate plan_a1.cpp's build_a1_il for it, the
// kernel instantiates them as compile-time constants (no descriptor loads, no kind / layout / flag branches, immediate
// LDS offsets), and plan.cpp sets A1Plan::p2_ok only if they reproduce the model's il_desc / il_fetch tables exactly.
namespace p2
{
constexpr int kLayers = 10, kJobs = 20, kDepth = 10, kXt = 3;
constexpr int kConstsB = 2 * kIlWinB;
constexpr int kXtB = kConstsB + kJobs * 256;
constexpr int kTilesB = kXtB + kXt * 1024;
constexpr int kFlagB = kTilesB + kJobs * kWsTileFloats * 4;
constexpr int kLdsBytes = kFlagB + 64;
constexpr int chan(int C0, int C1, int j) { return j < kLayers ? C0 : C1; }
constexpr int dil(int j) { return 1 << (j % kLayers); }
constexpr int kind(int j) { return dil(j) >= kBlock ? IL_HIST : dil(j) >= 4 ? IL_DPP : IL_EXCH; }
// float offset of job j's ring in the stream state: write-position table (64 words), then the rings in layer order
constexpr int ring_floats(int C0, int C1, int j)
{
  int off = kBlock;
  for (int k = 0; k < j; k++)
    off += chan(C0, C1, k) * (2 * dil(k) + kBlock);
  return off;
}
constexpr int xt_index(int j) { return j == 10 ? 1 : j == 19 ? 2 : 0; }
constexpr IlDesc desc(int C0, int C1, int act, int j)
{
  IlDesc d{};
  const int C = chan(C0, C1, j);
  d.flags = CD_LAYER | (C == 8 ? CD_HALF : 0) | (j == 0 ? CD_X0 : 0) | (j == 9 ? CD_POST_RECH : 0)
            | (j == 10 ? (CD_PRE_HEAD | (C0 == 8 ? CD_PREV_HALF : 0)) : 0) | (j == 19 ? CD_POST_OUT : 0);
  d.kind = kind(j);
  d.act = act;
  d.gp = 16 * (C / 4 - 1);
  d.ring_b = ring_floats(C0, C1, j) * 4;
  d.R = 2 * dil(j) + kBlock;
  d.ring_id = j;
  d.row_b = C * 4;
  d.dil = dil(j);
  d.tap0_lds = kind(j) == IL_EXCH ? 1 : 0;
  const int n = (j + 1) % kJobs;
  d.n_consts_b = kConstsB + n * 256;
  d.n_xt_b = kXtB + xt_index(n) * 1024;
  d.n_tiles_b = kTilesB + n * kWsTileFloats * 4;
  d.n_ready = n + 1;
  return d;
}
constexpr IlFetch fetch(int C0, int C1, int j)
{
  IlFetch f{};
  const int t = (j + kDepth) % kJobs; // the job the requests are for
  const int d = dil(t);
  f.ring_b = ring_floats(C0, C1, t) * 4;
  f.R = 2 * d + kBlock;
  f.ring_id = t;
  f.row_b = chan(C0, C1, t) * 4;
  f.nA = f.nB = 16;
  if (kind(t) == IL_HIST)
  {
    f.LA = 2 * d;
    f.LB = d;
  }
  else if (kind(t) == IL_DPP)
  {
    f.LA = 2 * d;
    f.LB = d;
    f.nA = d / 2 < 16 ? d / 2 : 16;
    f.nB = d / 4;
  }
  else
  {
    f.LA = kBlock;
    f.LB = 0;
  }
  return f;
}
} // namespace p2

// ---- K-tap MFMA kernel (nam_kt_mfma_kernel): single-array A1-family models with any per-layer kernel size ----
// (A2: K = 6 / 15, head rechannel K = 16.) A layer is cut into CHUNKS of up to kKtTaps taps; the current frame is a
// tap like any other (lookback 0). Only a layer's last chunk activates, applies the 1x1, publishes the layer output
// and meets the workgroup barrier; the head rechannel is one more "layer" whose input is the head accumulator.
// Same state (rings, write positions) and packed weights as nam_a1_kernel; the tiles below are an MFMA-operand
// repacking of those weights.
constexpr int kKtTaps = 6;
constexpr int kKtChunkMax = 112;
enum KtFlags : int32_t
{
  KT_FIRST = 1, // first chunk of a layer: accumulators start from bias + mixin * input sample; append the layer input to its ring
  KT_LAST = 2, // last chunk: activation, head accumulate, 1x1 + residual, publish, barrier
  KT_HEAD = 4, // the head rechannel (input = head accumulator; no activation / 1x1; LAST writes the output sample)
  KT_NEXT_HEAD = 8, // LAST of the final layer: publish the head accumulator instead of x
  KT_RING = 16 // the layer has a history ring (K > 1)
};
struct KtDesc // 16 x int32
{
  int32_t flags;
  int32_t ntaps; // taps in this chunk (1..kKtTaps)
  int32_t tile_off; // blob float offset of the chunk's tap tiles (plan_a1.cpp: build_a1_kt for the record layout)
  int32_t w1_off; // byte offset, inside the kernel's LDS copy, of the layer's 1x1 tile [64 lanes][NK]
  int32_t consts_off; // byte offset, same region, of the layer's constants in the lane layout: bias | mixin | 1x1 bias, 16 floats each
  int32_t ring_b; // byte offset of the layer's ring in the stream state
  int32_t R; // ring length in frames
  int32_t ring_id; // index into the write-position table (0 when the layer has no ring)
  int32_t L[kKtTaps]; // lookback of each tap in frames; unused slots: kKtNoTap
  int32_t pad[2];
};
constexpr int kKtNoTap = 1 << 20;
static_assert(sizeof(KtDesc) == 64, "descriptor size is part of the kernel ABI");

struct A1Plan
{
  int32_t valid = 0;
  int32_t n_arrays = 0;
  int32_t head_scale_off = 0; // blob offset
  int32_t n_rings = 0;
  int32_t pad[4] = {0, 0, 0, 0};
  int32_t ring_len_by_id[64]; // R of ring r (for the per-block write-position update)
  A1Array arr[kA1MaxArrays];
  // MFMA kernel
  int32_t ws_ok = 0; // nam_a1_mfma_kernel can run this model (plan_a1.cpp: build_a1_ws)
  int32_t ws_jobs = 0; // jobs per block (even)
  int32_t ws_tiles_off = 0, ws_consts_off = 0, ws_r1_off = 0; // blob offsets: tiles [jobs][1024], consts [jobs][64], 16 floats
  int32_t ws_xt_off = 0, ws_n_xt = 0; // blob offset / count of the extra tiles [n][256]
  int32_t ws_lds_tiles_b = 0, ws_lds_xt_b = 0, ws_lds_cond_b = 0, ws_lds_bytes = 0; // LDS layout (bytes)
  int32_t ws_prefetch = 6; // D
  CDesc cdesc[kWsJobMax];
  VDesc vdesc[kWsJobMax];
  // interleaved-frame MFMA kernel (shares ws_tiles_off / ws_consts_off / ws_xt_off with the kernel above)
  int32_t il_ok = 0;
  int32_t il_jobs = 0, il_real_jobs = 0; // jobs per block incl. padding / real layers
  int32_t il_depth = 5; // request depth = unroll factor (10 when it divides the layer count, else 5)
  int32_t il_exch = 0; // IL_EXCH jobs (= workgroup barriers) per block
  int32_t il_consts_b = 0, il_xt_b = 0, il_tiles_b = 0, il_flag_b = 0, il_lds_bytes = 0; // LDS layout (bytes); windows at 0
  IlDesc il_desc[kIlJobMax];
  IlFetch il_fetch[kIlJobMax];
  int32_t p2_ok = 0; // nam_a1_p2_kernel<p2_c0, p2_c1> runs this model (the official topology, see namespace p2)
  int32_t p2_c0 = 0, p2_c1 = 0; // the two arrays' (padded) channel counts
  // K-tap MFMA kernel
  int32_t kt_ok = 0; // nam_kt_mfma_kernel can run this model (plan_a1.cpp: build_a1_kt)
  int32_t kt_chunks = 0; // chunks per block
  int32_t kt_nk = 4; // k-steps per matrix: 2 = half layout (C = 8), 4 = full layout
  int32_t kt_rech_off = 0; // blob float offset: rechannel column in the lane layout (16 floats)
  int32_t kt_lds_src_off = 0, kt_lds_floats = 0; // blob region copied to LDS at kernel start: 1x1 tiles | constants
  KtDesc kt_desc[kKtChunkMax];
  int32_t kp_ok = 0; // nam_kq_kernel can run this model: it IS the topology of kp_table.h (plan_a1.cpp: build_a1_kp)
  int32_t kq_w_off = 0; // blob float offset of nam_kq_kernel's weight block (tiles | constants | rechannel column; kernel_kq.hip)
  int32_t q_ok = 0; // nam_a1_q_kernel runs this model: it IS the topology of aq_table.h (plan_a1.cpp: build_a1_q)
  int32_t q_w_off = 0; // blob float offset of its weight block (aq_table.h: kWrOff .. kBlockFloats)
};

// ---- register-resident WaveNet (nam_wn_reg_kernel) ----------------------------------------------------------------
// For WaveNets whose layers are a few channels wide (the FiLM-heavy A2 "max" feature set, nested condition_dsp included;
// the slimmable example's 1..3 channels with dilations up to 512): one wavefront per stream, lane = frame, EVERY
// activation in registers — a layer is one fully unrolled function instantiated per shape (kernel_wn_reg.hip:
// WR_SHAPES), run from a list of macro-ops (an array's rechannel, a layer, an array's head rechannel, "the nested net's
// output becomes the condition", "store the output"). The only LDS traffic is the weights (broadcast b128 reads) and each
// layer's conv input, which lives in an LDS-RESIDENT RING of exactly lookback + 64 frames (the reference's RingBuffer,
// NAM/ring_buffer.cpp:7-109, without the rewind): the dilation history never leaves the CU while a launch runs.
//   program: the macro-ops sit in the blob too (`tab_ops`): a global-memory fetch per op costs a cache round trip that
//            a small layer does not cover; from LDS the next op arrives while the current one runs
//   weights: `blob` is copied to LDS once per launch; per matrix [in][pad4(out)] (transposed), zero padded; grouped convs
//            expanded to dense (block-diagonal); a layer's block has a fixed layout given its shape (inactive FiLMs
//            keep their zeroed slot), see wr_layer_layout; behind the weights three int tables (below)
//   rings:   layer with C conv-input channels, kernel K, dilation d: R = (K - 1) d + 64 frames, stored as
//            [ceil(C / 4)][R][gs] floats (gs = 4, the last group C % 4): frame-major inside a group of up to four channels,
//            so a lane appends / fetches its frame's group with ONE ds_write / ds_read (b128 / b64 / b32: the lane stride
//            of every shape is bank-conflict free). One write position per layer ("slot").
//   state:   per stream [64 ints: the slots' write positions][the rings, exactly as in LDS]
//   LDS:     [blob][rings]; the write positions live in a register (lane = slot)
//   tables (int4 entries {float offset inside the ring area, R, slot | gs << 8, o}): entry = one 64-frame window of one
//            channel: lane j <-> ring index wrap(position - o + j), float offset + index * gs.
//            `rows`: one entry per channel with o = 0 (the block being written; the kernel substitutes o = frames
//            just processed when it stores them back); `pf`: the windows of history a single 64-frame block's taps can
//            reach (what a one-block launch fetches from the state instead of the whole ring); `ring`: R per slot.
enum WrOpType : int32_t
{
  WR_ARRAY_BEGIN = 0, // head accumulator = first array ? 0 : previous head output; x = rechannel(previous x | input)
  WR_LAYER = 1,
  WR_ARRAY_END = 2, // head output = head_rechannel(head accumulator) (+ bias)
  WR_SET_COND = 3, // condition registers = scale * head output (the nested condition_dsp's result)
  WR_OUTPUT = 4, // out[ch] = scale * head output[ch]
  // a run of consecutive PLAIN layers of one shape (condition size 1, bottleneck = channels <= 4, kernel size 3, no
  // gating / FiLM / head1x1, a parameterless activation): one dispatch for the run, the next layer's weights requested
  // while the current layer computes. `n_in` = layers, `w` = the first layer's weight block (the others follow at the
  // layout's stride), `hist` = LDS float offset of the per-layer records {unused, ring area offset, R, dilation | slot << 24}
  WR_RUN = 5,
  // head rechannel WITH TAPS (model.cpp:399-400, 547-548: a Conv1D of kernel size K_h over the head accumulator): the
  // head accumulator has a ring of its own (`hist`, `ring`, `dil`, `slot` as in WR_LAYER). Per-model compile only.
  WR_ARRAY_END_K = 6,
  WR_POST_HEAD = 7 // one layer of the post-stack head: activation, Conv1D (per-model compiled shapes only)
};

struct WrOp // 16 x int32 = 64 bytes
{
  int32_t type;
  int32_t shape; // index into the kernel's instantiation table (WR_LAYER / WR_ARRAY_*: (in, out) pair)
  int32_t w; // float offset of this op's weights in the blob (and in the LDS copy)
  int32_t hist; // WR_LAYER: LDS float offset of the layer's ring area
  int32_t ring; // WR_LAYER: R, frames per ring = (K - 1) * dilation + 64
  int32_t dil; // WR_LAYER: dilation
  int32_t flags; // WR_LAYER: bit i = FiLM slot i active, bit 8 + i = it has a shift; bit 16 = blended (else gated) when the
                 // shape is a gating one; WR_ARRAY_BEGIN: bit 0 = first array (head accumulator starts at 0, x from the
                 // input); WR_ARRAY_END: bit 0 = bias
  int32_t act, act2; // WR_LAYER: activation types (primary, secondary)
  int32_t n_in, n_out; // WR_ARRAY_BEGIN: input size, channels; WR_ARRAY_END: head input size, head size; SET_COND/OUTPUT: count
  float scale; // SET_COND / OUTPUT: head_scale
  int32_t slot; // WR_LAYER: index of the layer's write position
  int32_t run; // WR_LAYER (planner only): run shape id + 1 when the layer can join a WR_RUN, else 0
  int32_t pad[2];
};
static_assert(sizeof(WrOp) == 64, "WrOp must stay 64 bytes");

constexpr int kWrPosInts = 64; // write positions per stream (one per layer): the header of the state, a table in LDS
constexpr int kWrRegs = 8; // width of the register files (x, condition, head accumulator, head output)
constexpr int kWrQueueBytes = 5 * kWrRegs * 64 * 4 + 64; // two-stage launches: the registers in flight between the two waves + token / flag words
constexpr int kWrActFloats = 20; // per activation: p0..p3, then the PReLU slope of each of (up to) 16 rows
constexpr int kWrMaxLdsBytes = 156 * 1024; // a workgroup of nam_wn_reg_kernel may take (nearly) a whole CU's 160 KB of LDS
constexpr int kWrMaxGroups = 8; // width groups one launch of nam_wn_reg_kernel can serve (kernels.h: WrArgs)

struct WrPlan
{
  bool ok = false;
  std::string why; // when !ok: the first unsupported thing
  std::string jit_failed; // non-empty: the per-model compile this plan asked for was not available (why) — it was planned again without
  std::vector<WrOp> ops;
  std::vector<float> blob; // weights, then the tables (int bit patterns)
  int n_layers = 0; // = slots
  bool has_layers = false, has_runs = false; // op types in the program (which kernel instantiation can run it)
  bool has_rt_layers = false; // a WR_LAYER whose shape leaves FiLM set / activations to run-time flags
  int state_floats = 0; // per stream: kWrPosInts + hist_floats, rounded up to a multiple of 64
  int hist_floats = 0; // floats of the ring area (a multiple of 4)
  int lds_bytes = 0; // blob + positions + rings
  int tab_rows = 0, n_rows = 0; // blob float offset / entry count of the `rows` table
  int tab_pf = 0, n_pf = 0; // ... of the one-block prefetch windows
  int tab_ring = 0; // ... of R per slot (n_layers ints, padded to 4)
  int tab_ops = 0; // ... of the ops (16 ints each): the kernel reads its program from the LDS copy
  int program = -1; // per-model compile with the programs compiled in: this plan's program in the model's code object (WrShapeSet::programs)
  std::vector<std::array<int32_t, 4>> run_recs; // (planner) the records of this plan's WR_RUN layers, in op order; WrOp::pad[0] of a WR_RUN = its first
  int split_op[4] = {0, 0, 0, 0}; // pipelined launches: cuts of the program at 1/4, 1/2, 3/4 of its work (four wavefronts per stream), [3]: the two-wave cut (plan_wr.cpp: wr_program_cuts; kernel_wn_reg.hip, NST)
  // per-model compile (wr_jit.cpp): the op shapes are ids into the model's own WrShapeSet; `jit_module` is the code
  // object compiled for it ("" until api_launch.cpp: build_model has prepared it — the plan is not runnable before)
  bool jit = false;
  std::string jit_module;
};

// The layer shapes kernel_wn_reg.hip instantiates AHEAD OF TIME — (id, condition size, channels, bottleneck, gating,
// kernel size, head1x1 outputs (0 = no head1x1), FiLM mask, shift mask, blended, activation, secondary activation,
// layer1x1 active). A FiLM mask of -1 leaves the FiLM slots, the blend and the activation types to run-time flags
// (wavefront-uniform branches); a full description makes the layer one straight-line block the compiler can schedule
// weight reads across. The first match wins: exact descriptions first.
// Any OTHER model within the kernel's limits (<= 8 channels / condition rows / head1x1 outputs, <= 16 conv outputs,
// <= 156 KB of weights and rings) gets the same kernel compiled for exactly ITS shapes when it is loaded (wr_jit.cpp: the
// tables below are then generated per model, NAM_WR_JIT_SHAPES) — the way the reference templates its fast path on the
// channel count (NAM/wavenet/a2_fast.cpp:57), extended to every layer description.
#ifndef NAM_WR_JIT_SHAPES
#define WR_LAYER_SHAPES(X) \
  /* example_models/wavenet_a2_max.nam: main array; its condition_dsp's array 0 and the three layers of array 1 */ \
  X(0, 8, 4, 4, false, 4, 4, 0xff, 0xff, 0, ACT_SOFTSIGN, ACT_IDENTITY, 1) \
  X(1, 1, 3, 6, true, 2, 6, 0xff, 0xff, 0, ACT_SILU, ACT_HARDSWISH, 1) \
  X(2, 1, 4, 2, true, 3, 4, 0xff, 0x00, 1, ACT_PRELU, ACT_LEAKYHARDTANH, 1) \
  X(3, 1, 4, 2, true, 3, 4, 0xff, 0x00, 0, ACT_PRELU, ACT_RELU, 1) \
  X(4, 1, 4, 2, true, 3, 4, 0xff, 0x00, 0, ACT_SOFTSIGN, ACT_SIGMOID, 1) \
  /* example_models/slimmable_wavenet.nam at its three widths (plain ReLU layers, dilations 1 .. 512) */ \
  X(15, 1, 3, 3, false, 3, 0, 0x00, 0x00, 0, ACT_RELU, ACT_IDENTITY, 1) \
  X(16, 1, 2, 2, false, 3, 0, 0x00, 0x00, 0, ACT_RELU, ACT_IDENTITY, 1) \
  X(17, 1, 1, 1, false, 3, 0, 0x00, 0x00, 0, ACT_RELU, ACT_IDENTITY, 1) \
  /* run-time flags (used when the per-model compile is switched off or unavailable): the same shapes with other FiLM */ \
  /* sets / activations, plain small stacks (no head1x1), example_models/wavenet_condition_dsp.nam, multi-channel fixtures */ \
  X(5, 8, 4, 4, false, 4, 4, -1, 0, 0, -1, -1, 1) \
  X(6, 1, 3, 6, true, 2, 6, -1, 0, 0, -1, -1, 1) \
  X(7, 1, 4, 2, true, 3, 4, -1, 0, 0, -1, -1, 1) \
  X(8, 1, 4, 4, false, 3, 0, -1, 0, 0, -1, -1, 1) \
  X(9, 1, 3, 3, false, 3, 0, -1, 0, 0, -1, -1, 1) \
  X(10, 1, 2, 2, false, 3, 0, -1, 0, 0, -1, -1, 1) \
  X(11, 1, 8, 8, false, 3, 0, -1, 0, 0, -1, -1, 1) \
  X(12, 3, 3, 3, false, 3, 0, -1, 0, 0, -1, -1, 1) \
  X(13, 3, 4, 4, false, 3, 0, -1, 0, 0, -1, -1, 1) \
  X(14, 3, 2, 2, false, 3, 0, -1, 0, 0, -1, -1, 1) \
  X(18, 1, 1, 1, false, 3, 0, -1, 0, 0, -1, -1, 1)
// plain-layer runs (WR_RUN): (id, channels, activation)
#define WR_RUN_SHAPES(X) \
  X(0, 1, ACT_RELU) X(1, 2, ACT_RELU) X(2, 3, ACT_RELU) X(3, 4, ACT_RELU) \
  X(4, 1, ACT_TANH) X(5, 2, ACT_TANH) X(6, 3, ACT_TANH) X(7, 4, ACT_TANH) \
  X(8, 1, ACT_FASTTANH) X(9, 2, ACT_FASTTANH) X(10, 3, ACT_FASTTANH) X(11, 4, ACT_FASTTANH)
#define WR_PAIR_SHAPES(X) \
  X(0, 1, 3) X(1, 3, 4) X(2, 1, 4) X(3, 6, 4) X(4, 4, 8) X(5, 4, 1) X(6, 4, 4) X(7, 1, 2) X(8, 2, 1) X(9, 3, 1) X(10, 1, 8) \
  X(11, 8, 1) X(12, 8, 4) X(13, 4, 2) X(14, 2, 2) X(15, 3, 3) X(16, 8, 8) X(17, 2, 4) X(18, 3, 2) X(19, 6, 1) X(20, 1, 1) X(21, 2, 3) X(22, 4, 3) X(23, 3, 8) X(24, 2, 8)
// head rechannels with taps: (id, head accumulator rows, head size, kernel size) — none ahead of time
#define WR_HEADK_SHAPES(X)
// post-stack head layers (id, inputs, outputs, kernel size, activation): per-model compiled shapes only
#define WR_POSTHEAD_SHAPES(X)
#endif
// The shapes of one model (or of all the sub-models of a slimmable one), collected while it is planned, for the
// per-model compile: ids are positions in these lists.
struct WrShapeSet
{
  struct Layer
  {
    int cond, C, B, G, K, HO, flags, act, act2, l1;
  };
  struct Run
  {
    int C, act;
  };
  struct Pair
  {
    int n_in, n_out;
  };
  struct HeadK
  {
    int n_in, n_out, K;
  };
  std::vector<Layer> layers;
  std::vector<Run> runs;
  std::vector<Pair> pairs;
  std::vector<HeadK> heads; // head rechannels with kernel size > 1
  int head(int n_in, int n_out, int K);
  struct PostHead
  {
    int n_in, n_out, K, act;
  };
  std::vector<PostHead> posts; // post-stack head layers
  int post(int n_in, int n_out, int K, int act);
  int layer(int cond, int C, int B, bool G, int K, int HO, int flags, int act, int act2, bool l1);
  int run(int C, int act);
  int pair(int n_in, int n_out);
  // Round 6: the op PROGRAMS of the model's plans too (one per width / submodel): the per-model build compiles them in
  // (kernel_wn_reg.hip: NAM_WR_PROGRAMS) — no op fetch, no dispatch, every offset an immediate
  struct Program
  {
    std::vector<WrOp> ops; // as the blob holds them (LDS offsets final): what ONE wavefront per stream runs
    std::vector<WrOp> ops_cut; // ... and with every WR_RUN cut into up to four sub-runs: what two / four wavefronts per stream share
    int split_op[4]; // the cuts of ops_cut ([0 .. 2]: four waves, [3]: two)
    int first_rec; // WR_RUN ops: their layers' records start at run_recs[first_rec + (op.slot as stored here)]
  };
  std::vector<Program> programs;
  std::vector<std::array<int32_t, 4>> run_recs; // {w, ring area offset, R, dilation | slot << 24} of every layer of every WR_RUN
  bool empty() const { return layers.empty() && runs.empty() && pairs.empty() && heads.empty() && posts.empty(); }
  std::string header_text() const; // the generated tables: "#define NAM_WR_JIT_SHAPES 1 / #define WR_LAYER_SHAPES(X) ..."
};
// id of a shape in the ahead-of-time tables, -1 = not instantiated
int wr_layer_shape(int cond, int channels, int bottleneck, bool gating, int kernel, int head_out, int flags, int act,
                   int act2, bool l1 = true);
int wr_pair_shape(int n_in, int n_out);
bool wr_layer_shape_is_exact(int id); // FiLM set, blend and activations compiled in
int wr_run_shape(int channels, int act);
// float count / offsets of a layer's weight block (shared by the planner and the kernel)
struct WrLayerLayout
{
  int conv, conv_b, mixin, l1, l1_b, h1, h1_b, film[8], act, act2, total;
};
constexpr int wr_pad4(int n)
{
  return (n + 3) & ~3;
}
// a FiLM's two matrices in the MATRIX form (kernel_wn_reg.hip: WrFilm — [lane % 4][pad4(D) / 4][cond] for v_mfma_f32_4x4x1) instead
// of the vector form [cond][pad4(D)]: conditions of 4 or 8 values (what a condition_dsp hands to every FiLM of its model)
constexpr bool wr_film_matrix_form(int cond)
{
  return cond == 4 || cond == 8;
}
constexpr WrLayerLayout wr_layer_layout(int cond, int C, int B, bool gating, int K, int HO)
{
  // the conv (one matrix per tap), layer1x1 and head1x1 in the MATRIX form (kernel_wn_reg.hip: WrMatM): [lane % 4 = output row of
  // a quad][pad4(out) / 4 quads][pad4(in)] — the lane's own row, the A operand of v_mfma_f32_4x4x1 —; the mixin (and a FiLM on a
  // condition that is not 4 or 8 values) TRANSPOSED, [in][pad4(out)]: one b128 read = the weights of four outputs for one input,
  // i.e. two packed FMAs (v_pk_fma_f32: two outputs per instruction) with the input broadcast to both halves
  WrLayerLayout L{};
  const int zc = gating ? 2 * B : B;
  int o = 0;
  L.conv = o; // K x [4][pad4(zc) / 4][pad4(C)]
  o += K * wr_pad4(zc) * wr_pad4(C);
  L.conv_b = o;
  o += wr_pad4(zc);
  L.mixin = o; // [cond][pad4(zc)]
  o += cond * wr_pad4(zc);
  L.l1 = o; // [4][pad4(C) / 4][pad4(B)]
  o += wr_pad4(C) * wr_pad4(B);
  L.l1_b = o;
  o += wr_pad4(C);
  L.h1 = o; // [4][pad4(HO) / 4][pad4(B)]
  o += wr_pad4(HO) * wr_pad4(B);
  L.h1_b = o;
  o += wr_pad4(HO);
  const int dims[8] = {C, zc, cond, zc, zc, B, C, HO};
  for (int k = 0; k < 8; k++)
  {
    L.film[k] = o; // scale [cond][pad4(D)], shift [cond][pad4(D)], scale bias [pad4(D)], shift bias [pad4(D)]
    o += 2 * cond * wr_pad4(dims[k]) + 2 * wr_pad4(dims[k]);
  }
  L.act = o;
  o += kWrActFloats;
  L.act2 = o;
  o += kWrActFloats;
  L.total = o;
  return L;
}

// a PLAIN layer's weight block (WR_RUN): kernel size 3, C = bottleneck <= 4, condition size 1, nothing else. The two matrices
// in the MATRIX form (round 6; kernel_wn_reg.hip: wr_plain_layer runs them on v_mfma_f32_4x4x1, one lane per frame): conv
// [lane % 4 = output row][pad4(3 C) inputs, input = tap * C + channel], layer1x1 [output row][4 inputs]; conv bias, mixin and
// the 1x1's bias as rows of four floats (lane-uniform)
struct WrPlainLayout
{
  int conv, conv_b, mixin, l1, l1_b, total;
};
constexpr WrPlainLayout wr_plain_layout(int C)
{
  return WrPlainLayout{0, 4 * wr_pad4(3 * C), 4 * wr_pad4(3 * C) + 4, 4 * wr_pad4(3 * C) + 8, 4 * wr_pad4(3 * C) + 24, 4 * wr_pad4(3 * C) + 28};
}

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
  // nam_lstm_mfma_kernel (16 streams per wavefront, one v_mfma_f32_16x16x4_f32 per 4 units x 4 inputs):
  // one contiguous blob region [mf_off, mf_off + mf_floats) that the kernel copies to LDS verbatim:
  //   per layer l, per unit tile T (4 units), per k-step s: a 64-float A tile, lane (k = lane >> 4, i = lane & 15)
  //     = W[gate(i & 3) * H + 4T + (i >> 2)][column of input element 4s + k]   (rows permuted so that one lane
  //     group receives the i, f, g, o pre-activations of ONE unit); k-steps: ceil(I_l / 4) over the layer input,
  //     then ceil(H / 4) over the recurrent state;
  //   per layer, per tile: 16 bias floats [unit][gate]; head: ceil(H / 4) A tiles (row = output channel), 16 biases.
  int32_t mf_ok = 0;
  int32_t mf_off = 0, mf_floats = 0;
  int32_t mf_nt = 0; // unit tiles = ceil(H / 4)
  int32_t mf_layer_tiles[16] = {0}; // float offset (inside the region) of layer l's A tiles
  int32_t mf_layer_bias[16] = {0}; // ... of layer l's bias table [NT][16]
  int32_t mf_head_tiles = 0, mf_head_bias = 0;
  int32_t mf_lds_bytes = 0; // region + h (double buffered) + c + I/O tiles
};

struct Plan
{
  int arch = 0;
  int in_channels = 1, out_channels = 1;
  int prewarm_samples = 0;
  std::vector<NamOp> ops;
  std::vector<float> blob;
  int lds_rows = 0; // LDS rows (x 64 floats) the generic kernel needs per wavefront
  int generic_blob_floats = 0; // leading part of `blob` the op program reads (weights, biases, activation parameters)
  int n_rings = 0;
  int state_floats = 0; // per-stream state size (floats), multiple of 64; first n_rings words = write positions
  // true when the A1 kernels run a zero-padded copy of the model (plan_a1.cpp: pad_channels_for_mfma): their rings are
  // [R][C_padded] while the op program's are [R][C] — two state layouts, so switching between the generic kernel and
  // the A1 kernels needs freshly reset state (api_launch.cpp: launch_group enforces it)
  bool a1_padded_layout = false;
  A1Plan a1;
  LSTMPlan lstm;
  WrPlan wr;
  std::string describe() const;
};

// Build the plan for a (non-slimmable view of a) model. Throws std::runtime_error on unsupported shapes.
// `jit_shapes` (optional): where nam_wn_reg_kernel's plan may register layer shapes the ahead-of-time tables do not
// hold — the plan then has wr.jit set and needs the kernel compiled for that shape set.
Plan build_plan(const ModelSpec& model, WrShapeSet* jit_shapes = nullptr);
Plan build_wavenet_plan(const WaveNetSpec& wn, WrShapeSet* jit_shapes = nullptr);

} // namespace namhip
