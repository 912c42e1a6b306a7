// kp_table.h — the topology nam_kq_kernel (kernel_kq.hip) is compiled for, shared with the planner, which checks a model
// against it layer by layer before the kernel may run it (plan_a1.cpp: build_a1_kp).
//
// Default: the A2 architecture — one layer array of 8 channels, 23 layers with kernel sizes 6 / 15 and the dilation
// pattern below, a head rechannel with 16 taps; the shape the reference's own fused path is written for
// (NAM/wavenet/a2_fast.cpp:57-764 templates A2FastModel<C> on the channel count and hard-codes this stack; A2.nam's
// "A2-Full" submodel). Another single-array, 8-channel, K-tap topology can be compiled in by defining NAM_KP_TABLE with
// its own kLayers / kKs / kDs in front of this header.
#pragma once

namespace namhip
{
namespace kp
{
#ifndef NAM_KP_TABLE
constexpr int kLayers = 23; // conv layers; job kLayers is the head rechannel (a Conv1D over the head accumulator)
constexpr int kKs[kLayers + 1] = {6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 15, 15, 6, 6, 6, 6, 6, 6, 6, /*head*/ 16};
constexpr int kDs[kLayers + 1] = {1, 3, 7, 17, 41, 101, 239, 1, 3, 7, 17, 41, 101, 239, 1, 13, 1, 3, 7, 17, 41, 101, 239, /*head*/ 1};
#endif
constexpr int kC = 8; // channels (the half layout of the MFMA kernels: two k-steps per matrix)
constexpr int kJobs = kLayers + 1;
constexpr int kTapsPerChunk = 6; // == kKtTaps: the tap tiles are nam_kt_mfma_kernel's (plan_a1.cpp: build_a1_kt)
constexpr int kTable = 64; // write-position words in front of a stream's rings

constexpr int chunks_of(int K) { return (K + kTapsPerChunk - 1) / kTapsPerChunk; }
constexpr int chunk0(int job) // first chunk of a job in the K-tap kernel's chunk table
{
  int c = 0;
  for (int i = 0; i < job; i++)
    c += chunks_of(kKs[i]);
  return c;
}
constexpr int kChunks = chunk0(kJobs);
constexpr int ring_len(int job) { return (kKs[job] - 1) * kDs[job] + 64; }
constexpr int ring_off(int job) // float offset of the job's ring in the stream state (rings in job order behind the table)
{
  int o = kTable;
  for (int i = 0; i < job; i++)
    o += ring_len(i) * kC;
  return o;
}
constexpr int state_floats() { return ring_off(kJobs); }
// matrix instructions of a job per wave: two per tap, two for the layer's 1x1
constexpr int mfmas(int job) { return 2 * kKs[job] + (job < kLayers ? 2 : 0); }
constexpr int mfmas_before(int job)
{
  int m = 0;
  for (int i = 0; i < job; i++)
    m += mfmas(i);
  return m;
}
// first job of stage s of an nst-stage pipeline (stage nst = end): the cut closest to s / nst of the matrix work; every
// stage keeps at least three jobs (its ring requests run two jobs ahead) and the head job never starts a stage
constexpr int first_job(int nst, int s)
{
  if (s <= 0)
    return 0;
  if (s >= nst)
    return kJobs;
  const int total = mfmas_before(kJobs);
  int best = 3 * s, best_d = 1 << 30;
  for (int j = 3 * s; j <= kJobs - 3 * (nst - s) && j < kLayers; j++)
  {
    int d = mfmas_before(j) * nst - total * s;
    d = d < 0 ? -d : d;
    if (d < best_d)
    {
      best_d = d;
      best = j;
    }
  }
  return best;
}
constexpr int max_jobs(int nst)
{
  int m = 0;
  for (int s = 0; s < nst; s++)
    m = first_job(nst, s + 1) - first_job(nst, s) > m ? first_job(nst, s + 1) - first_job(nst, s) : m;
  return m;
}
constexpr int stage_of(int nst, int job)
{
  int s = 0;
  for (int k = 1; k < nst; k++)
    if (job >= first_job(nst, k))
      s = k;
  return s;
}
constexpr int max_k()
{
  int m = 0;
  for (int i = 0; i < kJobs; i++)
    m = kKs[i] > m ? kKs[i] : m;
  return m;
}
} // namespace kp
} // namespace namhip
