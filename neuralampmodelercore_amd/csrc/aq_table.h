// aq_table.h — the job / stage / LDS tables of nam_a1_q_kernel (kernel_a1_q.hip), shared with the planner (plan_a1.cpp:
// build_a1_q packs the weight block these offsets describe and checks the model against this topology).
//
// Topology: the official A1 "standard" WaveNet — two layer arrays of ten layers, kernel size 3, dilations 1 .. 512,
// 16 and 8 channels (NAM/wavenet/model.cpp:183-393, 463-549; example_models/wavenet_a1_standard.nam). 22 jobs per buffer:
//   0 .. 9   array 0's layers ("big": 16 channels; v_mfma_f32_16x16x4_f32, four groups of 16 frames per wavefront)
//   10       the transition: array 1's rechannel 16 -> 8 and array 0's head rechannel 16 -> 8 (one lane per frame, 4x4x1)
//   11 .. 20 array 1's layers ("small": 8 channels; one lane per frame, v_mfma_f32_4x4x1_16b_f32)
//   21       array 1's head rechannel 8 -> 1, head_scale, the output sample
// Ring r of the stream state (r = 0 .. 19) is the conv input of layer r: [R = 2 d + 64 frames][C channels], frame-major
// (the layout every A1 kernel shares: plan.h, namespace p2).
#pragma once

namespace namhip
{
namespace aq
{
constexpr int kC0 = 16, kC1 = 8;
constexpr int kBlockF = 64; // frames per buffer
constexpr int kLayers = 10; // per array
constexpr int kJobT = 10, kJobM0 = 11, kJobHead = 21, kJobs = 22;
constexpr int kRings = 20;
constexpr int kTable = 64; // write-position words in front of a stream's rings
constexpr int kNst = 16;
// stage s = jobs [kFirst[s], kFirst[s + 1]); wave i of the workgroup runs stage kStageOfWave[i]; waves i, i + 4, i + 8, i + 12
// share a SIMD (tools/src/simd_map.hip). FOUR waves per SIMD: one wave issues a vector instruction every ~8.6 cycles, three
// share the port at 3.0 cycles per instruction, four at ~2.2 (tools/src/valu_rate.hip) — and a stage's own instruction stream
// is a floor under the period whatever the SIMD's load (profiles/r04/a1q_timeline_12_stages.txt: the five-job last stage of
// the twelve-stage cut never waited). Every big layer is a stage of its own, the small layers go in pairs. Matrix + vector
// port cycles per buffer: a big layer 2.65 k, a pair of small layers 1.5 k -> per SIMD {L0, L1, L2, M9 + head} 8.85 k |
// {L3, L4, L5, T + M0} 9.25 k | {L6, L7, M1-2, M3-4} 8.3 k | {L8, L9, M5-6, M7-8} 8.35 k
constexpr int kFirst[kNst + 1] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 16, 18, 20, 22};
#ifdef NAM_AQ_SOW // (A/B builds: another stage -> wave table, profiles/r05/a1q_variants.txt)
constexpr int kStageOfWave[kNst] = {NAM_AQ_SOW};
#else
constexpr int kStageOfWave[kNst] = {0, 3, 6, 8, 1, 4, 7, 9, 2, 5, 11, 13, 15, 10, 12, 14};
#endif
static_assert(kFirst[kNst] == kJobs, "aq stage table");

constexpr bool is_big(int job) { return job < kLayers; }
constexpr bool is_small(int job) { return job >= kJobM0 && job < kJobHead; }
constexpr bool has_ring(int job) { return is_big(job) || is_small(job); }
constexpr int ring_id(int job) { return is_big(job) ? job : job - kJobM0 + kLayers; }
constexpr int dil(int job) { return has_ring(job) ? 1 << (is_big(job) ? job : job - kJobM0) : 0; }
constexpr int chans(int job) { return job <= kJobT ? kC0 : kC1; } // channels of the job's INPUT rows
constexpr int ring_len(int job) { return 2 * dil(job) + kBlockF; }
constexpr int ring_off(int job) // float offset of the job's ring in the stream state
{
  int o = kTable;
  for (int r = 0; r < ring_id(job); r++)
    o += (r < kLayers ? kC0 : kC1) * (2 * (1 << (r % kLayers)) + kBlockF);
  return o;
}
// LDS-resident rings: the whole ring lives in LDS while the launch runs (loaded from the state when it starts, written
// back when it leaves); the others keep their ring in HBM (appended every buffer, their taps — all at least two buffers
// old — requested one buffer ahead)
constexpr bool res(int job) { return is_big(job) ? dil(job) <= 64 : is_small(job) ? dil(job) <= 128 : false; }
constexpr int stage_of(int job)
{
  int s = 0;
  for (int k = 1; k < kNst; k++)
    if (job >= kFirst[k])
      s = k;
  return s;
}
constexpr bool starts_stage(int job) { return kFirst[stage_of(job)] == job; }
// a job's input area in LDS: planes of 16-byte rows, [C / 4 planes][rows]. Resident ring: R rows (plane pitch rounded up
// to 256 bytes: the four lane groups of a b128 access then never share a bank). A stage's FIRST job whose ring is in HBM
// (or that has none: the transition) takes its input through a small area: two sub-blocks (32 rows) for a big layer, one
// buffer (64 rows) for the transition and for a small layer.
constexpr bool takes_area(int job) { return job > 0 && job < kJobHead && starts_stage(job) && !res(job); }
// How many sub-blocks the hand-over INTO a big stage holds (its slot, and its input area when it has one): two — a resident ring of
// 2 d + 64 rows has room for the consumer's sub-block, its 2 d of history and at most three sub-blocks ahead of it, and the words
// count modulo a power of two — except into the stages whose first ring is in HBM (L7, L8, L9: they take their rows through an area
// of their own, which may be as deep as LDS allows): four with NAM_AQ_DEEP_Q (A/B: profiles/r05/a1q_variants.txt)
#if defined(NAM_AQ_DEPTH4_MASK) // (A/B builds: bit j = four sub-blocks into big job j)
constexpr int depth_in(int job) { return (is_big(job) && ((NAM_AQ_DEPTH4_MASK >> job) & 1)) ? 4 : 2; }
#elif defined(NAM_AQ_DEEP_Q)
constexpr int depth_in(int job) { return (is_big(job) && !res(job)) ? 4 : 2; }
#else
constexpr int depth_in(int) { return 2; }
#endif
constexpr int in_rows(int job) { return res(job) ? ring_len(job) : takes_area(job) ? (is_big(job) ? 16 * depth_in(job) : kBlockF) : 0; }
constexpr int plane_b(int job) { return (in_rows(job) * 16 + 255) / 256 * 256; }
constexpr int in_bytes(int job) { return plane_b(job) * (chans(job) / 4); }

// ---- weight block in the blob (floats), copied to LDS as it lies (plan_a1.cpp: build_a1_q) ----
// 4x4x1 tiles [lane class i = lane % 4][h][c] = W[out = 4 h + i][in = c]
constexpr int kTileT = 4 * 2 * 16; // 16 -> 8: 128 floats
constexpr int kTileM = 4 * 2 * 8; // 8 -> 8 (and the head's 8 -> 1: class 0, half 0 only): 64 floats
constexpr int kWrOff = 0; // array 1's rechannel
constexpr int kWhOff = kWrOff + kTileT; // array 0's head rechannel
constexpr int kMTiles = kWhOff + kTileT; // per small layer: tap 0 (oldest), 1, 2, 1x1
constexpr int kHeadTile = kMTiles + kLayers * 4 * kTileM;
constexpr int kMConsts = kHeadTile + kTileM; // per small layer: bias[8] | mixin[8] | 1x1 bias[8]
constexpr int kTConsts = kMConsts + kLayers * 24; // array 0's head-rechannel bias[8] | array 1's head bias, 0, 0, 0 | pad
constexpr int kBigConsts = kTConsts + 16; // per big layer: bias[16] | mixin[16] | 1x1 bias[16] | (job 0: rechannel column[16])
constexpr int kBlockFloats = kBigConsts + kLayers * 64;
static_assert(kBlockFloats % 4 == 0, "aq weight block");

// ---- LDS layout (bytes) ----
constexpr int kWB = 0;
constexpr int kFlagB = kWB + kBlockFloats * 4; // 256 bytes of single-writer words
constexpr int kSlotB0 = kFlagB + 256;
// boundary b = stage b -> b + 1: a slot with the head accumulator (planes of 16-byte rows), the input sample, a token.
// Into a big stage: two sub-blocks deep ([4 planes][32 rows] | input sample [32] | token); into the transition: a whole
// buffer of sub-blocks ([4 planes][64 rows] | [64] | token); between small stages: one buffer ([2 planes][64 rows] | [64] | token)
constexpr int kSubSlot = 2 * 1024 + 128 + 16, kBigSlot = 4 * 1024 + 256 + 16, kSmallSlot = 2 * 1024 + 256 + 16;
constexpr int slot_bytes(int b) { return is_big(kFirst[b + 1]) ? (depth_in(kFirst[b + 1]) == 4 ? kBigSlot : kSubSlot) : kFirst[b + 1] == kJobT ? kBigSlot : kSmallSlot; }
constexpr int slot_b(int b)
{
  int o = kSlotB0;
  for (int i = 0; i < b; i++)
    o += slot_bytes(i);
  return o;
}
constexpr int kInB0 = (slot_b(kNst - 1) + 255) / 256 * 256;
constexpr int in_b(int job)
{
  int o = kInB0;
  for (int i = 0; i < job; i++)
    o += in_bytes(i);
  return o;
}
constexpr int kLdsBytes = in_b(kJobs);
static_assert(kLdsBytes <= 160 * 1024, "aq LDS layout");
static_assert(kFlagB % 16 == 0 && kSlotB0 % 16 == 0 && kSubSlot % 16 == 0 && kBigSlot % 16 == 0 && kSmallSlot % 16 == 0, "aq LDS alignment");
} // namespace aq
} // namespace namhip
