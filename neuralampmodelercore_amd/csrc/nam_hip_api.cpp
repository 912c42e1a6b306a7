// nam_hip_api.cpp — implementation of the C ABI declared in include/nam_hip.h.
//
// Host-side runtime around the HIP kernels: model handles (parsed .nam + device plans), batch
// handles (device weights, per-stream history in HBM, staging), stream sharding by slimmable width,
// reset / prewarm semantics of nam::DSP (NAM/dsp.cpp:67-140). No exception leaves this file.
#include "../../include/nam_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "kernels.h"
#include "model_spec.h"
#include "plan.h"
#include "wr_jit.h"

using namespace namhip;

namespace namhip
{
thread_local hipEvent_t tl_session_stop_event = nullptr; // (kernels.h: nam_launch)
}

namespace
{

thread_local std::string g_last_error;

int fail(int code, const std::string& msg)
{
  g_last_error = msg;
  return code;
}

#define NAM_HIP_CHECK(expr)                                                                                           \
  do                                                                                                                   \
  {                                                                                                                    \
    hipError_t _e = (expr);                                                                                            \
    if (_e != hipSuccess)                                                                                              \
      return fail(NAM_HIP_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));                             \
  } while (0)

template <typename F>
int guarded(F&& f)
{
  try
  {
    return f();
  }
  catch (const FileValidationError& e)
  {
    return fail(NAM_HIP_ERR_FILE, e.what());
  }
  catch (const std::exception& e)
  {
    return fail(NAM_HIP_ERR_MODEL, e.what());
  }
  catch (...)
  {
    return fail(NAM_HIP_ERR_MODEL, "unknown error");
  }
}
} // namespace

struct nam_hip_model
{
  std::shared_ptr<ModelSpec> spec;
  // One plan per distinct width (slimmable WaveNets have several; everything else exactly one).
  std::vector<std::vector<int>> width_channels;
  std::vector<Plan> plans;
  int full_width = 0; // index of the full-size plan

  bool slimmable() const { return spec->arch == ARCH_CONTAINER || (spec->arch == ARCH_WAVENET && spec->wavenet.slimmable); }
  int width_for_ratio(double ratio) const
  {
    if (spec->arch == ARCH_CONTAINER)
      return spec->container_index(ratio); // plan i = submodel i
    if (!spec->wavenet.slimmable || spec->arch != ARCH_WAVENET)
      return 0;
    const std::vector<int> ch = channels_for_ratio(spec->wavenet, ratio);
    for (size_t i = 0; i < width_channels.size(); i++)
      if (width_channels[i] == ch)
        return (int)i;
    return -1;
  }
};

namespace
{
struct WidthGroup
{
  const Plan* plan = nullptr;
  float* d_blob = nullptr;
  NamOp* d_ops = nullptr;
  A1Plan* d_a1 = nullptr;
  float* d_wr_blob = nullptr; // nam_wn_reg_kernel's weights, tables and macro-ops (plan.h: WrPlan)
  float* d_state = nullptr; // [n_streams][state_stride] (allocated when the first stream joins)
  float* d_init = nullptr; // LSTM initial state
  float* d_scratch = nullptr; // LSTM cells too large for LDS: nam_lstm_kernel<true>'s h / c / gate columns
  long scratch_floats = 0;
  long state_stride = 0;
  std::vector<int> streams; // members, ascending
  int* d_map = nullptr; // device copy of `streams` (nullptr when the group is all streams in order)
  // Which layout the state currently holds: -1 = freshly zeroed (any), 0 = the op program's rings (shared by the A1
  // kernels unless they run on zero-padded channels: plan.h, Plan::a1_padded_layout), 1 = the padded A1 rings,
  // 2 = nam_wn_reg_kernel's 64-frame conv-input histories
  int state_family = -1;
  // Prewarm cache (the reference caches what prewarm leaves in every conv, conv1d.cpp:151-161 / model.cpp:737-775, and
  // later Resets refill from it): one stream's state right after zero + prewarm — every stream's is the same — keyed by
  // the kernel that produced it and the frames it ran. A later Reset / SetSlimmableSize copies it instead of running
  // the silence again.
  float* d_prewarm = nullptr;
  int prewarm_kernel = -1, prewarm_len = 0;
};
} // namespace

namespace
{
// Persistent block mode (nam_hip_batch_set_persistent): one resident launch of nam_a1_p2_kernel per session, fed one
// command per 64-frame buffer through a device-memory ring (kernel_a1_p2.hip, PERSIST).
constexpr unsigned kPRing = 1024; // commands in flight at most (power of two)
// behind the ring: d_ring[kPRing] = the "leave" word of sessions whose launch lingers (ticketed host buffers: A1Args::p_linger) —
// the host stores the session's command count there when it wants the launch gone (a flush, the end of the session): a
// workgroup that has consumed exactly that many commands and finds no next one leaves at once instead of lingering
constexpr unsigned kPRingTail = 8;
constexpr int kTicketLingerDefault = 20000; // 200 us of the 100 MHz clock: workgroups drift apart by up to NAM_HIP_PIPE_SLOTS buffers (16 x 5.3 us) —
                                     // the one in front must outwait the host, which hands the next buffer in when the LAST one has finished an old one
// (NAM_HIP_TICKET_LINGER_US overrides it; 0 or 1: a ticket session's launch leaves as promptly as any other — for hosts that run several
// sessions on one device, where a lingering launch of one holds the CUs the other's launch is waiting for)
inline int ticket_linger_from_env() // (read when a batch is created, like the other switches)
{
  const char* e = std::getenv("NAM_HIP_TICKET_LINGER_US");
  return e ? (int)std::min(std::max(std::atol(e), 0l), 100000l) * 100 : kTicketLingerDefault;
}
struct PersistSession
{
  bool enabled = false; // the caller opted in
  bool active = false; // a window is registered; a launch of the session may be consuming commands
  // the command ring: (seq << 32) | frame offset, in FINE-GRAINED device memory — local to the workgroups that poll
  // it, and host-writable through the PCIe BAR (MI355X exposes all of HBM): the host stores a command itself when the
  // caller's stream is idle (the usual real-time case: nothing to order behind; a posted write, ~0.1 us), else the
  // store is enqueued on that stream (hipStreamWriteValue64: ~4 us of host time and a small kernel on the device)
  unsigned long long* d_ring = nullptr;
  bool host_store_ok = false; // the ring is fine-grained memory (else plain device memory: stream-ordered stores only)
  unsigned* h_words = nullptr; // host-mapped: [0, n_wg) progress, [n_wg, 2 n_wg) completion (bit 31 = exited)
  unsigned* d_words = nullptr; // the same words as the device sees them
  unsigned* d_cons = nullptr; // device memory: commands consumed per workgroup (where its next launch resumes)
  unsigned* d_cmd_count = nullptr; // device memory [kPRing]: workgroups through command c (A1Args::p_cmd_count), zero between commands
  unsigned *h_cmd_done = nullptr, *d_cmd_done = nullptr; // host-mapped [kPRing]: c + 1 once every workgroup is through command c
  hipStream_t last_caller = nullptr; // the stream the last doorbell was rung on
  int grace = 0; // A1Args::p_grace of the next launch
  long long seq0 = -1; // A1Args::p_seq0 / p_cmd0 of the next launch
  unsigned long long cmd0 = 0;
  unsigned flushed = 0; // every workgroup has consumed exactly this many commands (valid while == seq)
  bool flushed_valid = false;
  bool outstanding = false; // a launch of the session may still be running
  bool need_order = false; // the next launch must wait for the batch's own stream (session start)
  hipStream_t kstream = nullptr; // the resident launch's own stream (nothing else may be enqueued behind it)
  hipEvent_t order = nullptr; // makes the launch wait for what the caller had enqueued before the first buffer
  unsigned seq = 0; // commands submitted in this session
  const float* in_base = nullptr;
  float* out_base = nullptr;
  bool out_is_host = false; // the output window is host memory (A1Args::p_out_host)
  long stride = 0;
  int n_wg = 0; // workgroups of the session's launch
  int kind = -1; // PersistKind
  int done_off = 0; // h_words: [0, done_off) progress words, [done_off, 2 done_off) completion words
  // Sequence numbers are 31-bit (bit 31 of a completion word is the "left" flag): a session START — where every
  // workgroup stands at exactly `seq` and nothing is in flight — rebases them to 0 once they pass this mark
  // (NAM_HIP_PERSIST_REBASE_AT overrides it: tests)
  unsigned rebase_at = 0x40000000u;
  bool prepared = false; // persist_prepare ran to its end (every window-independent resource is there)
  bool rebase_pending = false; // persist_submit ended the session because the next buffer would cross the rebase mark: persist_start renumbers
  long timeout_ms = 20000; // a resident launch that makes no progress for this long is a device failure (NAM_HIP_PERSIST_TIMEOUT_MS)
  // developer statistics (NAM_HIP_SESSION_STATS=1: printed when the batch is destroyed)
  unsigned long long n_launches = 0, n_host_doorbells = 0, n_stream_doorbells = 0, n_starts = 0, n_flush_relaunches = 0;
  double t_poll = 0, t_out = 0, t_in = 0, t_cmd = 0, t_poll_max = 0, t_out_max = 0, t_in_max = 0, t_cmd_max = 0; // us (NAM_HIP_SESSION_STATS)
  long long *h_why = nullptr, *d_why = nullptr; // (NAM_HIP_SESSION_STATS) per workgroup: reason << 56 | grace loop << 48 | all-through count << 24 | own count
  unsigned long long n_waits = 0, n_polls = 0; // ticket waits, looks at the buffer's completion word
  unsigned epoch = 0; // counts session starts (a ticket of an earlier session is complete: sessions end flushed)
  // Burst lengths (commands between two whole flushes) of this session, newest first; ~0u = not seen yet. A host that flushes after
  // every buffer or two (a device-resident real-time chain: process_device + flush per 64 .. 256 frames) waits for the FIRST buffer of
  // every launch: the official 16 / 8 topology then starts as nam_a1_p4_kernel (four waves per layer: the first buffer is through in
  // ~6 us) instead of nam_a1_q_kernel (one wave per layer: ~27 us, faster only once buffers overlap) — the rule of the blocking host
  // calls (short_blocking_call), learnt from the caller's own pattern: three bursts in a row of at most four buffers
  unsigned bursts[3] = {~0u, ~0u, ~0u};
  unsigned burst_start = 0; // `seq` at the last whole flush
  bool short_bursts() const { return bursts[0] <= 4u && bursts[1] <= 4u && bursts[2] <= 4u; }
  bool one_buffer_bursts() const { return bursts[0] == 1u && bursts[1] == 1u && bursts[2] == 1u; } // (nam_wn_reg_kernel: one wave per stream then)
  hipEvent_t retired = nullptr; // the completion signal of the session's latest launch (kernels.h: nam_launch), recorded by the dispatch itself
  bool cmd_done_published = false; // the running launch stores p_cmd_done behind every command's results (A1Args::p_out_host == 2); p_prog stays ring bookkeeping every 16 commands
};

// One buffer in flight between nam_hip_batch_submit_f32 and nam_hip_batch_wait_f32
struct PipeSlot
{
  long long ticket = -1;
  bool in_flight = false;
  int n_frames = 0;
  int how = 0; // 0: a command range of the host-mapped session | 1: copies + launch on the batch's stream, `done` behind them | 2: rendered by a blocking call, kept in `held`
  unsigned seq_end = 0, epoch = 0; // how == 0: the session's command count behind this buffer, the session it belongs to
  hipEvent_t done = nullptr;
  std::vector<float> held;
};
} // namespace

struct nam_hip_batch
{
  const nam_hip_model* model = nullptr;
  int device = 0;
  int n_streams = 0;
  int max_frames = 0;
  hipStream_t stream = nullptr;
  std::vector<WidthGroup> groups;
  std::vector<int> stream_width;
  float* d_in = nullptr; // staging for the host-pointer entry points
  float* d_out = nullptr;
  float* h_stage = nullptr; // pinned, used by the f64 path
  // host-mapped staging of the blocking entry points in persistent mode: the session's kernel reads the input from and
  // writes the output to host memory itself (its input loads / output stores are system-scope anyway), so a blocking
  // call is: copy in, store the command(s), watch the completion words, copy out — no launch of a copy, no stream sync
  // input: FINE-GRAINED DEVICE memory the host writes through the PCIe BAR (posted writes; the device then reads local
  // HBM — device reads of host memory serialise at a microsecond or two per wavefront: 1.9 ms per buffer at 256 streams);
  // output: host-mapped memory the device writes (posted writes again), read by the host from its own DRAM
  float* in_bar = nullptr; // one address for both sides
  float *h_out_map = nullptr, *d_out_map = nullptr; // host address / the same memory as the device sees it
  bool map_failed = false; // the allocation was refused once: the copying path stays
  // the ticketed entry points (nam_hip_batch_submit_f32) have windows of their own, the same two kinds of memory,
  // NAM_HIP_PIPE_SLOTS buffers deep: [slot][row][max_frames], slot = ticket % NAM_HIP_PIPE_SLOTS
  float* pipe_in_bar = nullptr;
  float *pipe_h_out_map = nullptr, *pipe_d_out_map = nullptr;
  bool pipe_map_failed = false;
  PipeSlot pipe[NAM_HIP_PIPE_SLOTS];
  long long pipe_next = 0; // the next ticket
  bool pipe_session = false; // the session serves ticketed buffers: its launches publish every command (PersistSession::cmd_done_published)
  float *pipe_h_in = nullptr, *pipe_h_out = nullptr, *pipe_d_in = nullptr, *pipe_d_out = nullptr; // staging of the copying form ([slot][row][max_frames])
  std::vector<float> pipe_cvt; // the _f64 forms of submit / wait: one buffer of float32 on the way in / out
  int kernel = NAM_HIP_KERNEL_AUTO;
  long long* dbg = nullptr; // device buffer of the profiling instantiation (nam_hip_batch_debug_timeline)
  bool was_reset = false;
  bool reset_with_prewarm = true; // thread_local gPrewarmOnResetDefault = true (NAM/dsp.cpp:20)
  // the caller-supplied stream of the last nam_hip_batch_process_device: control calls that free or rewrite device
  // memory (Reset, SetSlimmableSize, destroy) wait for it as well as for the batch's own stream
  hipStream_t last_ext_stream = nullptr;
  bool short_blocking_call = false; // a blocking host call of up to four buffers is being served: the caller waits for it, so the FIRST buffer's
                                    // latency is what counts — nam_a1_p4_kernel (four waves per layer: ~6 us through the model) rather than
                                    // nam_a1_q_kernel (one wave per layer: ~30 us; faster only once buffers overlap)
  int wr_last_stages = 0; // (NAM_HIP_SESSION_STATS: what nam_wn_reg_kernel's last multi-buffer launch ran as)
  bool wr_last_dense = false;
  bool blocking_linger = false; // blocking host calls are coming back to back (the previous one returned < kBlockingLingerGapUs ago): the session's
                                // launch publishes every command and lingers for the next call, like a ticket session's
  double t_blocking_return = -1e18; // host clock (us) when the last blocking host call of the session path returned
  bool one_buffer_call = false; // a blocking host call of ONE 64-frame buffer is being served: nothing to overlap, a launch started now runs nam_wn_reg_kernel as one wave per stream
  int blocking_linger_us = 200; // 0 = blocking host calls never make a launch linger (NAM_HIP_BLOCKING_LINGER_US)
  int blocking_linger_gap_us = 50; // "back to back": the previous blocking call returned less than this ago
  int ticket_linger = kTicketLingerDefault; // ticks of the 100 MHz clock a ticket session's launch looks for the next buffer (NAM_HIP_TICKET_LINGER_US)
  // NAM_HIP_MAX_STAGES = 1 / 2 / 4 (developer switch; default: no cap): the most pipeline stages a stream is spread over.
  // 1 = no pipelines at all (`no_pipe`: nam_a1_p2_kernel where nam_a1_p4 / q would run, nam_kt_mfma_kernel instead of nam_kq_kernel,
  // nam_wn_reg_kernel as one wavefront per stream — the A/B and reference renderings of the tests); 2 / 4 cap nam_wn_reg_kernel's
  // wavefronts per stream (the compile-time pipelines have fixed stage counts)
  int wr_max_stages = 4;
  bool no_pipe = false;
  PersistSession ps;
  bool ps_launching = false; // launch_group is starting the session's resident launch
  int n_cus = 0; // compute units of the device
};

namespace
{

int upload_group(nam_hip_batch* b, WidthGroup& g)
{
  const Plan& p = *g.plan;
  NAM_HIP_CHECK(hipMalloc(&g.d_blob, std::max<size_t>(p.blob.size(), 1) * sizeof(float)));
  if (!p.blob.empty())
    NAM_HIP_CHECK(hipMemcpy(g.d_blob, p.blob.data(), p.blob.size() * sizeof(float), hipMemcpyHostToDevice));
  if (p.arch == ARCH_WAVENET)
  {
    NAM_HIP_CHECK(hipMalloc(&g.d_ops, p.ops.size() * sizeof(NamOp)));
    NAM_HIP_CHECK(hipMemcpy(g.d_ops, p.ops.data(), p.ops.size() * sizeof(NamOp), hipMemcpyHostToDevice));
    if (p.a1.valid)
    {
      NAM_HIP_CHECK(hipMalloc(&g.d_a1, sizeof(A1Plan)));
      NAM_HIP_CHECK(hipMemcpy(g.d_a1, &p.a1, sizeof(A1Plan), hipMemcpyHostToDevice));
    }
    if (p.wr.ok)
    {
      NAM_HIP_CHECK(hipMalloc(&g.d_wr_blob, p.wr.blob.size() * sizeof(float)));
      NAM_HIP_CHECK(hipMemcpy(g.d_wr_blob, p.wr.blob.data(), p.wr.blob.size() * sizeof(float), hipMemcpyHostToDevice));
    }
  }
  else if (p.arch == ARCH_LSTM)
  {
    const auto& init = p.lstm.init_state;
    NAM_HIP_CHECK(hipMalloc(&g.d_init, std::max<size_t>(init.size(), 1) * sizeof(float)));
    NAM_HIP_CHECK(hipMemcpy(g.d_init, init.data(), init.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  g.state_stride = p.state_floats;
  (void)b;
  return NAM_HIP_OK;
}

int ensure_state(nam_hip_batch* b, WidthGroup& g)
{
  if (g.d_state)
    return NAM_HIP_OK;
  const size_t bytes = (size_t)b->n_streams * g.state_stride * sizeof(float);
  NAM_HIP_CHECK(hipMalloc(&g.d_state, bytes));
  NAM_HIP_CHECK(hipMemsetAsync(g.d_state, 0, bytes, b->stream));
  if (g.plan->arch == ARCH_LSTM) // h0 / c0 from the weight stream, once per (sub)model instance (lstm.cpp:24-28)
    NAM_HIP_CHECK(launch_fill_state(g.d_state, g.state_stride, nullptr, b->n_streams, g.d_init,
                                    (int)g.plan->lstm.init_state.size(), g.plan->state_floats, b->stream));
  return NAM_HIP_OK;
}

// Wait for everything the batch may still have in flight: its own stream and the last caller-supplied one.
int persist_stop(nam_hip_batch* b);

hipError_t quiesce(nam_hip_batch* b)
{
  if (b->ps.active && persist_stop(b) != NAM_HIP_OK) // a resident launch owns the streams' state until it has left
    return hipErrorUnknown;
  hipError_t e = b->stream ? hipStreamSynchronize(b->stream) : hipSuccess;
  if (b->last_ext_stream && b->last_ext_stream != b->stream)
  {
    const hipError_t e2 = hipStreamSynchronize(b->last_ext_stream);
    if (e == hipSuccess)
      e = e2;
  }
  return e;
}

int state_family_of(const Plan& p, int kernel)
{
  if (kernel == NAM_HIP_KERNEL_WN_REG)
    return 2;
  return (p.a1_padded_layout && kernel != NAM_HIP_KERNEL_GENERIC) ? 1 : 0;
}

int refresh_map(nam_hip_batch* b, WidthGroup& g)
{
  if (g.d_map)
  {
    NAM_HIP_CHECK(quiesce(b));
    NAM_HIP_CHECK(hipFree(g.d_map));
    g.d_map = nullptr;
  }
  bool identity = (int)g.streams.size() == b->n_streams;
  for (size_t i = 0; identity && i < g.streams.size(); i++)
    identity = g.streams[i] == (int)i;
  if (g.streams.empty() || identity)
    return NAM_HIP_OK;
  NAM_HIP_CHECK(hipMalloc(&g.d_map, g.streams.size() * sizeof(int)));
  NAM_HIP_CHECK(hipMemcpy(g.d_map, g.streams.data(), g.streams.size() * sizeof(int), hipMemcpyHostToDevice));
  return NAM_HIP_OK;
}

// Which kernel a WaveNet group runs: explicit choice if possible, otherwise the fastest available.
constexpr size_t kKtAutoMaxStreams = 1024;

constexpr int kPersistTurns = 8; // sessions whose workgroups cannot all be on the chip at once: up to this many turns (they consume the same commands one after the other)
int pick_kernel(const nam_hip_batch* b, const WidthGroup& g)
{
  const bool a1 = g.plan->a1.valid && g.d_a1;
  const bool mfma = a1 && (g.plan->a1.ws_ok || g.plan->a1.kt_ok);
  const bool il = a1 && g.plan->a1.il_ok && g.plan->a1.p2_ok; // the interleaved-frame kernels: the official topologies (compile-time job tables)
  const bool wr = g.plan->wr.ok && g.d_wr_blob;
  // a model no A1 kernel takes (FiLMs, gating, a nested condition_dsp ...) runs with its activations in registers when
  // its layers are among the instantiated shapes, else through the op interpreter
  const int fallback = a1 ? NAM_HIP_KERNEL_A1 : (wr ? NAM_HIP_KERNEL_WN_REG : NAM_HIP_KERNEL_GENERIC);
  switch (b->kernel)
  {
    case NAM_HIP_KERNEL_GENERIC: return NAM_HIP_KERNEL_GENERIC;
    case NAM_HIP_KERNEL_WN_REG: return wr ? NAM_HIP_KERNEL_WN_REG : fallback;
    case NAM_HIP_KERNEL_A1: return fallback;
    case NAM_HIP_KERNEL_A1_MFMA: return mfma ? NAM_HIP_KERNEL_A1_MFMA : fallback;
    case NAM_HIP_KERNEL_A1_IL: return il ? NAM_HIP_KERNEL_A1_IL : (mfma ? NAM_HIP_KERNEL_A1_MFMA : fallback);
    default: // AUTO
      // narrow models (1 .. 8 channels in the instantiated layer shapes) keep their whole dilation history in LDS on
      // nam_wn_reg_kernel; the VALU kernel fetches it from the HBM rings layer by layer
      if (!mfma)
      {
        // ... as long as the batch fits the chip that way (LDS image x streams per CU): beyond it no session can hold the
        // batch (its workgroups may take turns on the chip: kPersistTurns) and every buffer is a launch that moves the
        // image's windows in and out — a plain model then runs its HBM rings on the VALU kernel (A2-Lite, 105 KB of rings
        // per stream: 8.1 k xRT at any stream count with a launch per buffer, 13.2 k / 21.7 k / 40.5 k at 512 / 1,024 / 2,048
        // streams on the VALU kernel; 48.6 k in a session at 256).
        // Decided on the batch's stream count, which never changes: the two kernels keep different state layouts.
        if (wr && a1)
        {
          const int per_cu = std::min(4, (160 * 1024) / (g.plan->wr.lds_bytes + 512));
          if (b->n_streams > kPersistTurns * std::max(per_cu, 1) * std::max(b->n_cus, 1)) // (a session's workgroups may take turns)
            return NAM_HIP_KERNEL_A1;
        }
        return wr ? NAM_HIP_KERNEL_WN_REG : fallback;
      }
      // The K-tap kernel (A2 shapes) spreads a stream over four wavefronts: 2.3x the VALU kernel while the chip has
      // idle SIMDs, level with it at ~1,000 streams per GPU, behind it beyond (it issues more instructions per tap).
      if (!g.plan->a1.ws_ok && g.streams.size() > kKtAutoMaxStreams)
        return fallback;
      return NAM_HIP_KERNEL_A1_MFMA;
  }
}

// Name of the __global__ function launch_group runs for this group (what rocprofv3 --kernel-trace reports, without
// template arguments): lets callers attribute measurements to the right kernel.
enum PersistKind : int
{
  PERSIST_NONE = -1,
  PERSIST_A1_P2 = 0, // nam_a1_q_kernel / nam_a1_p4_kernel (nam_a1_p2_kernel with NAM_HIP_MAX_STAGES=1): one workgroup (most of a CU's LDS) per stream
  PERSIST_WN_REG = 1, // nam_wn_reg_kernel: one wavefront per stream
  PERSIST_LSTM_ROW = 2, // nam_lstm_row_kernel: one wavefront per four streams
  PERSIST_LSTM_WIDE = 3, // nam_lstm_wide_kernel: one wavefront per stream
  PERSIST_KQ = 4 // nam_kq_kernel (the A2 topology): one workgroup (most of a CU's LDS) per stream, as PERSIST_A1_P2
};
int persist_kind(const nam_hip_batch* b);
int persist_family(const nam_hip_batch* b, const WidthGroup& g);
inline bool persist_eligible(const nam_hip_batch* b)
{
  return persist_kind(b) != PERSIST_NONE;
}

int kernel_for_launch(const nam_hip_batch* b, const WidthGroup& g, int n_frames);
// nam_a1_p4_kernel (the official topology as a pipeline of wave sets, consecutive buffers in flight at once) instead of
// nam_a1_p2_kernel: whenever a launch holds more than one buffer — a persistent session, an offline render, a prewarm. A
// launch of one block has nothing to overlap (every stage waits for the one before) and keeps the four-wave kernel.
inline bool use_pipeline(const nam_hip_batch* b, int n_frames)
{
  return !b->no_pipe && (b->ps_launching || n_frames > kBlock);
}

// the official 16 / 8 topology's pipeline: nam_a1_q_kernel (one-wave stages, LDS-resident rings) for the activations it is compiled
// for, nam_a1_p4_kernel otherwise (and for the other official sizes)
inline bool q_runs(const nam_hip_batch*, const Plan& p)
{
  return p.a1.q_ok && a1_q_takes(p.a1.arr[0].act);
}

// the A2 topology's pipeline: nam_kq_kernel (one lane per frame, 4x4x1 matrix instructions) for the activations it is compiled
// for (kernel_kq.hip: kq_takes); any other activation on that topology has no pipeline: nam_kt_mfma_kernel, a launch per buffer
inline bool kq_runs(const nam_hip_batch*, const Plan& p)
{
  return p.a1.kp_ok && kq_takes(p.a1.arr[0].act, p.a1.arr[0].act_p0);
}

// how long a session's launch that publishes every command looks for the next one (ticks of the 100 MHz clock)
inline int session_linger_ticks(const nam_hip_batch* b)
{
  return (b->blocking_linger && !b->pipe_session) ? b->blocking_linger_us * 100 : b->ticket_linger;
}

// `n_frames`: the launch length the question is about (under AUTO a launch of four or more blocks runs another kernel
// of the family than a one-block launch); 64 in persistent mode means "a command of the session"
const char* group_kernel_name(const nam_hip_batch* b, const WidthGroup& g, int n_frames)
{
  const Plan& p = *g.plan;
  if (b->ps.enabled && n_frames == kBlock)
    switch (persist_kind(b)) // persistent block mode
    {
      case PERSIST_A1_P2: // (what the NEXT launch of the session starts: PersistSession::short_bursts)
        return b->no_pipe ? "nam_a1_p2_kernel" : (q_runs(b, p) && !(!b->pipe_session && b->ps.short_bursts())) ? "nam_a1_q_kernel" : "nam_a1_p4_kernel"; // (launch_group's own predicate: the burst history outlives a session)
      case PERSIST_KQ: return "nam_kq_kernel";
      case PERSIST_WN_REG: return "nam_wn_reg_kernel";
      case PERSIST_LSTM_ROW: return "nam_lstm_row_kernel";
      case PERSIST_LSTM_WIDE: return "nam_lstm_wide_kernel";
      default: break;
    }
  if (p.arch == ARCH_WAVENET)
  {
    switch (kernel_for_launch(b, g, n_frames))
    {
      case NAM_HIP_KERNEL_GENERIC: return "nam_generic_kernel";
      case NAM_HIP_KERNEL_WN_REG: return "nam_wn_reg_kernel";
      case NAM_HIP_KERNEL_A1: return "nam_a1_kernel";
      case NAM_HIP_KERNEL_A1_IL:
        return (!b->no_pipe && n_frames > kBlock) ? (q_runs(b, p) ? "nam_a1_q_kernel" : "nam_a1_p4_kernel") : "nam_a1_p2_kernel";
      default:
        return p.a1.ws_ok ? "nam_a1_mfma_kernel" : (kq_runs(b, p) && !b->no_pipe && n_frames > kBlock) ? "nam_kq_kernel" : "nam_kt_mfma_kernel";
    }
  }
  const LSTMPlan& L = p.lstm;
  if (b->kernel != NAM_HIP_KERNEL_GENERIC && b->kernel != NAM_HIP_KERNEL_A1_MFMA && L.hidden >= 1 && L.hidden <= 4
      && L.n_layers <= 2 && L.input_size >= 1 && L.input_size <= 2 && L.in_ch == L.input_size && L.out_ch <= 16)
    return "nam_lstm_row_kernel";
  if (b->kernel != NAM_HIP_KERNEL_GENERIC && b->kernel != NAM_HIP_KERNEL_A1_MFMA && L.hidden >= 5 && L.hidden <= 32
      && L.n_layers <= 2 && L.input_size >= 1 && L.input_size <= 2 && L.in_ch == L.input_size && L.out_ch <= 16)
    return "nam_lstm_wide_kernel";
  if (L.mf_ok && b->kernel != NAM_HIP_KERNEL_GENERIC)
    return (L.input_size <= 4 && L.n_layers <= 2 && L.mf_nt <= 6) ? "nam_lstm_mfma_reg_kernel" : "nam_lstm_mfma_kernel";
  return "nam_lstm_kernel";
}

inline double stat_now_us()
{
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline bool stats_on()
{
  static const bool on = [] { const char* e = std::getenv("NAM_HIP_SESSION_STATS"); return e && e[0] == '1'; }();
  return on;
}

// the session's side of a persistent launch of a one-wavefront-per-workgroup kernel (kernels.h: PersistArgs)
PersistArgs persist_args(const nam_hip_batch* b)
{
  PersistArgs pa;
  if (!b->ps_launching)
    return pa;
  pa.ring = b->ps.d_ring;
  pa.ring_mask = (int)kPRing - 1;
  pa.cons = b->ps.d_cons;
  pa.prog = b->ps.d_words;
  pa.done = b->ps.d_words + b->ps.done_off;
  pa.seq0 = b->ps.seq0;
  pa.cmd0 = b->ps.cmd0;
  pa.grace = b->ps.grace;
  return pa;
}

// The function of a per-model code object on the current device (hipModuleLoad is per device: cached per path and
// device for the life of the process; a handful of entries).
// `dense`: the form built for two wavefronts per SIMD (kernel_wn_reg.hip: nam_wn_reg_jit2d / 4d); *dense_ok (optional) reports
// which stage counts have one the compiler fitted into 256 registers WITHOUT scratch (bit 1: two stages, bit 2: four).
int wr_jit_function(const std::string& path, int device, int stages, void** fn, bool dense = false, int* dense_ok = nullptr)
{
  struct Entry
  {
    std::string path;
    int device;
    hipModule_t module;
    hipFunction_t fn, fn2, fn4; // nam_wn_reg_jit, nam_wn_reg_jit2 (two stages), nam_wn_reg_jit4
    hipFunction_t fn2d, fn4d; // the dense forms (nullptr: not usable)
  };
  static std::vector<Entry> cache;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  auto pick = [&](const Entry& e) {
    if (dense_ok)
      *dense_ok = (e.fn2d ? 2 : 0) | (e.fn4d ? 4 : 0);
    if (fn)
      *fn = reinterpret_cast<void*>(stages == 4 ? (dense && e.fn4d ? e.fn4d : e.fn4) : stages == 2 ? (dense && e.fn2d ? e.fn2d : e.fn2) : e.fn);
  };
  for (const Entry& e : cache)
    if (e.device == device && e.path == path)
    {
      pick(e);
      return NAM_HIP_OK;
    }
  Entry e{path, device, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  NAM_HIP_CHECK(hipModuleLoad(&e.module, path.c_str()));
  NAM_HIP_CHECK(hipModuleGetFunction(&e.fn, e.module, "nam_wn_reg_jit"));
  NAM_HIP_CHECK(hipModuleGetFunction(&e.fn2, e.module, "nam_wn_reg_jit2"));
  NAM_HIP_CHECK(hipModuleGetFunction(&e.fn4, e.module, "nam_wn_reg_jit4"));
  // more than the default 64 KB of dynamic LDS per workgroup (long dilations: up to 156 KB of rings)
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(e.fn), hipFuncAttributeMaxDynamicSharedMemorySize, kWrMaxLdsBytes);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(e.fn2), hipFuncAttributeMaxDynamicSharedMemorySize, kWrMaxLdsBytes);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(e.fn4), hipFuncAttributeMaxDynamicSharedMemorySize, kWrMaxLdsBytes);
  static const bool dense_on = [] { const char* v = std::getenv("NAM_HIP_WR_DENSE"); return !(v && v[0] == '0'); }();
  for (int q = 0; q < 2 && dense_on; q++)
  {
    hipFunction_t f = nullptr;
    if (hipModuleGetFunction(&f, e.module, q == 0 ? "nam_wn_reg_jit2d" : "nam_wn_reg_jit4d") != hipSuccess || !f)
      continue;
    int scratch = 1, regs = 1 << 20;
    if (hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, f) != hipSuccess
        || hipFuncGetAttribute(&regs, HIP_FUNC_ATTRIBUTE_NUM_REGS, f) != hipSuccess || scratch > 32 || regs > 256)
      continue; // (spilled more than a handful of registers, or not a two-per-SIMD build after all: the one-wave-per-SIMD form serves)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, kWrMaxLdsBytes);
    (q == 0 ? e.fn2d : e.fn4d) = f;
  }
  (void)hipGetLastError();
  cache.push_back(e);
  pick(e);
  return NAM_HIP_OK;
}

// nam_wn_reg_kernel over up to kWrMaxGroups width groups in ONE launch (kernels.h: WrArgs): group k's `counts[k]` streams
// (`maps[k]`: position -> stream index, nullptr = identity) become consecutive workgroups.
int launch_wr(nam_hip_batch* b, WidthGroup* const* groups, const int* const* maps, const int* counts, int n_groups,
              const float* d_in, float* d_out, int n_frames, long io_stride, hipStream_t s)
{
  WrArgs a;
  std::memset(&a, 0, sizeof(a));
  int total = 0, lds_bytes = 0;
  bool layers = false, runs = false, rt_layers = false, can_split = true, can_split4 = true;
  for (int k = 0; k < n_groups; k++)
  {
    WidthGroup& g = *groups[k];
    const WrPlan& w = g.plan->wr;
    layers = layers || w.has_layers;
    runs = runs || w.has_runs;
    rt_layers = rt_layers || w.has_rt_layers;
    if (g.state_family >= 0 && g.state_family != 2)
      return fail(NAM_HIP_ERR_INVALID_ARGUMENT,
                  "kernel change crosses state layouts (the op program's rings, the A1 kernels' zero-padded rings and "
                  "nam_wn_reg_kernel's LDS-image rings differ): call nam_hip_batch_reset before switching");
    g.state_family = 2;
    WrGroup& G = a.g[k];
    G.blob = g.d_wr_blob;
    G.state = g.d_state;
    G.stream_map = maps[k];
    G.state_stride = g.state_stride;
    G.n_ops = (int)w.ops.size();
    G.blob_floats = (int)w.blob.size();
    G.hist_floats = w.hist_floats;
    G.n_slots = w.n_layers;
    G.tab_rows = w.tab_rows;
    G.n_rows = w.n_rows;
    G.tab_pf = w.tab_pf;
    G.n_pf = w.n_pf;
    G.tab_ring = w.tab_ring;
    G.tab_ops = w.tab_ops;
    G.first = total;
    for (int q = 0; q < 3; q++)
      G.split_op[q] = w.split_op[q];
    G.prog = w.program;
    can_split = can_split && w.split_op[1] >= 1 && w.split_op[1] < (int)w.ops.size();
    can_split4 = can_split4 && w.split_op[0] >= 1 && w.split_op[0] < w.split_op[1] && w.split_op[1] < w.split_op[2]
                 && w.split_op[2] < (int)w.ops.size();
    total += counts[k];
    lds_bytes = std::max(lds_bytes, w.lds_bytes);
  }
  // Two or four wavefronts per stream (the program cut up, consecutive buffers in flight: kernel_wn_reg.hip, NST) when the
  // launch holds more than one buffer and the chip has the SIMDs for it — config 4's 512 streams become 1,024
  // wavefronts, 256 streams too
  int stages = 1;
  bool dense = false; // the two-wavefronts-per-SIMD build of the per-model code object
  // (a session whose caller flushes after EVERY buffer is a series of one-buffer calls: nothing for a pipeline to overlap)
  const bool one_buffer_bursts = b->ps_launching && !b->pipe_session && b->ps.one_buffer_bursts();
  if (!b->no_pipe && !b->one_buffer_call && !one_buffer_bursts && (b->ps_launching || n_frames > kBlock))
  {
    // the launch's relative duration with nst waves per stream: a workgroup is nst waves at one wave per SIMD plus its LDS
    // image (and the queues), the workgroups beyond what the chip holds run in later turns (persist_kind), and a stream's
    // buffer takes 1, 1/1.75, 1/3.1 of the one-wave time (measured: DESIGN 4.5)
    const int cus = std::max(b->n_cus, 1);
    auto duration = [&](int nst) {
      const int lds = lds_bytes + (nst - 1) * kWrQueueBytes;
      if (lds > kWrMaxLdsBytes)
        return 1e9;
      const int on_chip = cus * std::min(4 / nst, (160 * 1024) / (lds + 512));
      const double speed = nst == 4 ? 3.1 : nst == 2 ? 1.75 : 1.0;
      const int turns = (total + on_chip - 1) / on_chip;
      return turns * (1.0 + 0.15 * (turns - 1)) / speed; // a turn's last workgroups leave SIMDs idle; images move in and out
    };
    double best = duration(1);
    if (can_split && b->wr_max_stages >= 2 && duration(2) < 0.9 * best)
    {
      stages = 2;
      best = duration(2);
    }
    if (can_split && can_split4 && b->wr_max_stages >= 4 && duration(4) < 0.9 * best)
    {
      stages = 4;
      best = duration(4);
    }
    // ... or the DENSE forms of a per-model code object (two wavefronts per SIMD: twice the workgroups per CU; two waves that
    // share a SIMD each issue nearly as fast as a lone one — profiles/r05/valu_rate_microbench.txt: 8.6 cycles per instruction of
    // a wave at one AND at two per SIMD — minus what they lose to each other's LDS traffic: 0.9)
    const std::string& module0 = groups[0]->plan->wr.jit_module;
    if (!module0.empty() && can_split && b->wr_max_stages >= 2)
    {
      int ok = 0;
      if (wr_jit_function(module0, b->device, 1, nullptr, false, &ok) == NAM_HIP_OK && ok != 0)
      {
        auto duration_dense = [&](int nst) {
          const int lds = lds_bytes + (nst - 1) * kWrQueueBytes;
          if (lds > kWrMaxLdsBytes)
            return 1e9;
          const int on_chip = cus * std::min(8 / nst, (160 * 1024) / (lds + 512));
          const double speed = 0.9 * (nst == 4 ? 3.1 : 1.75);
          const int turns = (total + on_chip - 1) / on_chip;
          return turns * (1.0 + 0.15 * (turns - 1)) / speed;
        };
        if ((ok & 2) && duration_dense(2) < 0.9 * best)
        {
          stages = 2;
          dense = true;
          best = duration_dense(2);
        }
        if ((ok & 4) && can_split4 && b->wr_max_stages >= 4 && duration_dense(4) < 0.9 * best)
        {
          stages = 4;
          dense = true;
        }
      }
    }
    lds_bytes += (stages - 1) * kWrQueueBytes;
    if (stats_on() && (stages != b->wr_last_stages || dense != b->wr_last_dense))
      std::fprintf(stderr, "nam_hip nam_wn_reg_kernel: %d workgroups as %d wavefront(s) per stream%s, %d bytes of LDS each\n", total, stages,
                   dense ? " (two per SIMD)" : "", lds_bytes);
    b->wr_last_stages = stages;
    b->wr_last_dense = dense;
  }
  a.n_groups = n_groups;
  a.in = d_in;
  a.out = d_out;
  a.io_stride = io_stride;
  a.n_frames = n_frames;
  a.in_ch = groups[0]->plan->in_channels;
  a.out_ch = groups[0]->plan->out_channels;
  a.ps = persist_args(b);
  // every group on the model's own code object (they share one: build_model), or every group on the ahead-of-time kernel
  const std::string& module = groups[0]->plan->wr.jit_module;
  for (int k = 1; k < n_groups; k++)
    if (groups[k]->plan->wr.jit_module != module)
      return fail(NAM_HIP_ERR_INVALID_ARGUMENT, "nam_wn_reg_kernel: the groups of one launch run different code objects");
  if (!module.empty())
  {
    void* fn = nullptr;
    const int rc = wr_jit_function(module, b->device, stages, &fn, dense);
    if (rc != NAM_HIP_OK)
      return rc;
    NAM_HIP_CHECK(launch_wn_reg_jit(fn, a, total, lds_bytes, stages, s));
    return NAM_HIP_OK;
  }
  NAM_HIP_CHECK(launch_wn_reg(a, total, lds_bytes, layers, runs, rt_layers, stages, s));
  return NAM_HIP_OK;
}

// The non-empty groups of a batch when ALL of them run nam_wn_reg_kernel (then one launch serves the whole batch, and a
// persistent session can too); n = 0 otherwise. (A fixed array: this runs inside process calls, which allocate nothing.)
struct WrGroupList
{
  WidthGroup* g[kWrMaxGroups];
  int n = 0;
};
WrGroupList wr_groups(nam_hip_batch* b)
{
  WrGroupList out;
  const Plan& full = *b->groups[b->model->full_width].plan;
  for (auto& g : b->groups)
  {
    if (g.streams.empty())
      continue;
    if (out.n == kWrMaxGroups || g.plan->arch != ARCH_WAVENET || !g.d_wr_blob || pick_kernel(b, g) != NAM_HIP_KERNEL_WN_REG
        || g.plan->in_channels != full.in_channels || g.plan->out_channels != full.out_channels
        || (out.n > 0 && g.plan->wr.jit_module != out.g[0]->plan->wr.jit_module)) // (one launch = one code object)
    {
      out.n = 0;
      return out;
    }
    out.g[out.n++] = &g;
  }
  return out;
}
int launch_wr_all(nam_hip_batch* b, const WrGroupList& gs, const float* d_in, float* d_out, int n_frames, long io_stride,
                  hipStream_t s)
{
  const int* maps[kWrMaxGroups];
  int counts[kWrMaxGroups];
  for (int k = 0; k < gs.n; k++)
  {
    maps[k] = gs.g[k]->d_map;
    counts[k] = (int)gs.g[k]->streams.size();
  }
  return launch_wr(b, gs.g, maps, counts, gs.n, d_in, d_out, n_frames, io_stride, s);
}

// The WaveNet kernel a launch of n_frames runs: pick_kernel, except that under AUTO a launch that walks several blocks
// (offline render, prewarm) takes the interleaved-frame kernel — the faster one inside a launch (9.3 vs 11.2 us per
// block at 256 streams; its longer prologue only hurts one-block launches). Same rings, same write positions: the two
// alternate freely.
int kernel_for_launch(const nam_hip_batch* b, const WidthGroup& g, int n_frames)
{
  const int kernel = pick_kernel(b, g);
  if (b->kernel == NAM_HIP_KERNEL_AUTO && kernel == NAM_HIP_KERNEL_A1_MFMA && g.plan->a1.ws_ok && g.plan->a1.il_ok && g.plan->a1.p2_ok
      && n_frames >= 4 * kBlock)
    return NAM_HIP_KERNEL_A1_IL;
  return kernel;
}

// Launch one group's kernel over `n` streams given by `d_map` (nullptr = streams 0..n-1).
int launch_group(nam_hip_batch* b, WidthGroup& g, const int* d_map, int n, const float* d_in, float* d_out,
                 int n_frames, long io_stride, hipStream_t s)
{
  if (n <= 0 || n_frames <= 0)
    return NAM_HIP_OK;
  const Plan& p = *g.plan;
  if (p.arch == ARCH_WAVENET)
  {
    const int kernel = kernel_for_launch(b, g, n_frames);
    // the op program and the A1 kernels of a channel-padded model keep different ring layouts: a change of kernel
    // family is only legal on freshly reset state
    const int fam = state_family_of(p, kernel);
    if (g.state_family >= 0 && g.state_family != fam)
      return fail(NAM_HIP_ERR_INVALID_ARGUMENT,
                  "kernel change crosses state layouts (the op program's rings, the A1 kernels' zero-padded rings and "
                  "nam_wn_reg_kernel's conv-input histories differ): call nam_hip_batch_reset before switching");
    g.state_family = fam;
    if (kernel == NAM_HIP_KERNEL_WN_REG)
    {
      WidthGroup* one[1] = {&g};
      const int* maps[1] = {d_map};
      const int counts[1] = {n};
      return launch_wr(b, one, maps, counts, 1, d_in, d_out, n_frames, io_stride, s);
    }
    if (kernel != NAM_HIP_KERNEL_GENERIC)
    {
      A1Args a;
      a.plan = g.d_a1;
      a.blob = g.d_blob;
      a.state = g.d_state;
      a.state_stride = g.state_stride;
      a.stream_map = d_map;
      a.in = d_in;
      a.out = d_out;
      a.io_stride = io_stride;
      a.n_frames = n_frames;
      a.act_p0 = p.a1.arr[0].act_p0; // uniform across arrays and layers for the A1 kernels (plan.cpp)
      a.dbg = b->dbg;
      if (b->ps_launching && b->ps.h_why)
        a.dbg = b->ps.d_why; // (NAM_HIP_SESSION_STATS: why a lingering workgroup left — il_common.h: session_wait_command)
      a.n_rings = p.a1.n_rings;
      a.head_scale = p.blob[(size_t)p.a1.head_scale_off];
      a.n_mjobs = a.tiles_off = a.consts_off = 0;
      a.r1_off = a.xt_off = a.n_xt = a.lds_tiles_b = a.lds_xt_b = a.lds_cond_b = a.lds_bytes = a.prefetch = 0;
      a.il_jobs = a.il_real_jobs = a.il_depth = a.il_exch = 0;
      a.il_consts_b = a.il_xt_b = a.il_tiles_b = a.il_flag_b = a.il_lds_bytes = a.act = 0;
      a.p_ring = nullptr;
      a.p_ring_mask = 0;
      a.p_cons = a.p_prog = a.p_done = nullptr;
      a.p_grace = 0;
      a.p_out_host = 0;
      a.p_linger = 0;
      a.p_cmd_count = a.p_cmd_done = nullptr;
      a.p_seq0 = -1;
      a.p_cmd0 = 0;
      if (kernel == NAM_HIP_KERNEL_A1_IL)
      {
        int act = p.a1.arr[0].act;
        for (int i = 1; i < p.a1.n_arrays; i++)
          if (p.a1.arr[i].act != act)
            act = -1;
        a.tiles_off = p.a1.ws_tiles_off;
        a.consts_off = p.a1.ws_consts_off;
        a.xt_off = p.a1.ws_xt_off;
        a.n_xt = p.a1.ws_n_xt;
        a.il_jobs = p.a1.il_jobs;
        a.il_real_jobs = p.a1.il_real_jobs;
        a.il_depth = p.a1.il_depth;
        a.il_exch = p.a1.il_exch;
        a.il_consts_b = p.a1.il_consts_b;
        a.il_xt_b = p.a1.il_xt_b;
        a.il_tiles_b = p.a1.il_tiles_b;
        a.il_flag_b = p.a1.il_flag_b;
        a.il_lds_bytes = p.a1.il_lds_bytes;
        a.act = act;
        if (b->ps_launching)
        {
          a.p_ring = b->ps.d_ring;
          a.p_ring_mask = (int)kPRing - 1;
          a.p_cons = b->ps.d_cons;
          a.p_prog = b->ps.d_words;
          a.p_done = b->ps.d_words + b->ps.done_off;
          a.p_grace = b->ps.grace;
          a.p_out_host = b->ps.out_is_host ? (b->ps.cmd_done_published ? 2 : 1) : 0;
          a.p_linger = (b->ps.cmd_done_published && b->ps.host_store_ok && b->ps.n_wg <= b->n_cus) ? session_linger_ticks(b) : 0; // (more workgroups than CUs take turns: the ones on the chip must leave when the ring is empty)
          a.p_cmd_count = b->ps.d_cmd_count;
          a.p_cmd_done = b->ps.d_cmd_done;
          a.p_seq0 = b->ps.seq0;
          a.p_cmd0 = b->ps.cmd0;
        }
        if (!p.a1.p2_ok) // (pick_kernel: the interleaved-frame kernels exist for the official topologies' compile-time tables only)
          return fail(NAM_HIP_ERR_UNSUPPORTED, "NAM_HIP_KERNEL_A1_IL: not one of the official topologies");
        if (use_pipeline(b, n_frames) && q_runs(b, p) && !b->short_blocking_call && !(b->ps_launching && !b->pipe_session && b->ps.short_bursts()))
        {
          // the 16 / 8 topology as twelve one-wave stages, most rings resident in LDS (kernel_a1_q.hip): its own weight block
          // + the FULL-layout tiles of array 0 (kept in registers)
          a.consts_off = p.a1.ws_tiles_off;
          a.tiles_off = p.a1.q_w_off;
          NAM_HIP_CHECK(launch_a1_q(a, n, act, s));
        }
        else if (use_pipeline(b, n_frames))
          // ... as a pipeline of wave sets (three wavefronts per SIMD) across consecutive buffers
          NAM_HIP_CHECK(launch_a1_p4(a, n, p.a1.p2_c0, p.a1.p2_c1, act, s));
        else // one buffer: the four-wave kernel, job table compiled in
          NAM_HIP_CHECK(launch_a1_p2(a, n, p.a1.p2_c0, p.a1.p2_c1, act, s));
      }
      else if (kernel == NAM_HIP_KERNEL_A1_MFMA && !p.a1.ws_ok && n_frames > (1 << 28))
        // the K-tap kernel addresses the launch's input through a 32-bit buffer descriptor (1 GiB of float32 audio per
        // stream and launch): longer launches take the VALU kernel, same state layout
        NAM_HIP_CHECK(launch_a1(a, n, s));
      else if (kernel == NAM_HIP_KERNEL_A1_MFMA && !p.a1.ws_ok && kq_runs(b, p) && use_pipeline(b, n_frames))
      {
        // the A2 topology with more than one buffer in the launch (a session, a render, a prewarm): the pipeline of one-wave
        // stages compiled for it (kernel_kq.hip); same state as the K-tap kernel below
        a.tiles_off = p.a1.kt_desc[0].tile_off;
        a.consts_off = p.a1.kt_lds_src_off;
        a.r1_off = p.a1.kt_rech_off;
        a.act = p.a1.arr[0].act;
        if (b->ps_launching)
        {
          a.p_ring = b->ps.d_ring;
          a.p_ring_mask = (int)kPRing - 1;
          a.p_cons = b->ps.d_cons;
          a.p_prog = b->ps.d_words;
          a.p_done = b->ps.d_words + b->ps.done_off;
          a.p_grace = b->ps.grace;
          a.p_out_host = b->ps.out_is_host ? (b->ps.cmd_done_published ? 2 : 1) : 0;
          a.p_linger = (b->ps.cmd_done_published && b->ps.host_store_ok && b->ps.n_wg <= b->n_cus) ? session_linger_ticks(b) : 0; // (more workgroups than CUs take turns: the ones on the chip must leave when the ring is empty)
          a.p_cmd_count = b->ps.d_cmd_count;
          a.p_cmd_done = b->ps.d_cmd_done;
          a.p_seq0 = b->ps.seq0;
          a.p_cmd0 = b->ps.cmd0;
        }
        a.tiles_off = p.a1.kq_w_off;
        NAM_HIP_CHECK(launch_kq(a, n, p.a1.arr[0].act, s));
      }
      else if (kernel == NAM_HIP_KERNEL_A1_MFMA && !p.a1.ws_ok)
        // single-array models with other kernel sizes than 3 (A2): the K-tap MFMA kernel
        NAM_HIP_CHECK(launch_kt_mfma(a, n, p.a1.kt_nk, p.a1.arr[0].channels, p.a1.kt_lds_floats, p.a1.arr[0].act, s));
      else if (kernel == NAM_HIP_KERNEL_A1_MFMA)
      {
        // uniform activation across arrays -> compile-time specialised kernel, else run-time dispatch
        int act = p.a1.arr[0].act;
        for (int i = 1; i < p.a1.n_arrays; i++)
          if (p.a1.arr[i].act != act)
            act = -1;
        a.n_mjobs = p.a1.ws_jobs;
        a.tiles_off = p.a1.ws_tiles_off;
        a.consts_off = p.a1.ws_consts_off;
        a.r1_off = p.a1.ws_r1_off;
        a.xt_off = p.a1.ws_xt_off;
        a.n_xt = p.a1.ws_n_xt;
        a.lds_tiles_b = p.a1.ws_lds_tiles_b;
        a.lds_xt_b = p.a1.ws_lds_xt_b;
        a.lds_cond_b = p.a1.ws_lds_cond_b;
        a.lds_bytes = p.a1.ws_lds_bytes;
        a.prefetch = p.a1.ws_prefetch;
        NAM_HIP_CHECK(launch_a1_mfma(a, n, act, s));
      }
      else
        NAM_HIP_CHECK(launch_a1(a, n, s));
    }
    else
    {
      GenericArgs a;
      a.ops = g.d_ops;
      a.blob = g.d_blob;
      a.state = g.d_state;
      a.state_stride = g.state_stride;
      a.stream_map = d_map;
      a.in = d_in;
      a.out = d_out;
      a.io_stride = io_stride;
      a.n_frames = n_frames;
      a.in_ch = p.in_channels;
      a.out_ch = p.out_channels;
      // conv weights from LDS when the model's weights fit next to the activation rows (kernels.h)
      int lds_bytes = p.lds_rows * kBlock * (int)sizeof(float);
      a.w_lds_off = p.lds_rows * kBlock;
      a.blob_floats = 0;
      if (lds_bytes + p.generic_blob_floats * (int)sizeof(float) <= 96 * 1024)
      {
        a.blob_floats = p.generic_blob_floats;
        lds_bytes += p.generic_blob_floats * (int)sizeof(float);
      }
      NAM_HIP_CHECK(launch_generic(a, n, lds_bytes, s));
    }
  }
  else
  {
    const LSTMPlan& L = p.lstm;
    LSTMArgs a;
    a.blob = g.d_blob;
    a.state = g.d_state;
    a.state_stride = g.state_stride;
    a.stream_map = d_map;
    a.in = d_in;
    a.out = d_out;
    a.io_stride = io_stride;
    a.n_frames = n_frames;
    a.n_streams = n;
    a.n_layers = L.n_layers;
    a.input_size = L.input_size;
    a.hidden = L.hidden;
    a.in_ch = L.in_ch;
    a.out_ch = L.out_ch;
    a.fast = L.fast;
    a.head_w = L.head_w;
    a.head_b = L.head_b;
    for (int i = 0; i < 16; i++)
    {
      a.layer_w[i] = L.layer_w[i];
      a.layer_b[i] = L.layer_b[i];
    }
    a.mf_off = L.mf_off;
    a.mf_floats = L.mf_floats;
    a.mf_nt = L.mf_nt;
    a.mf_head_tiles = L.mf_head_tiles;
    a.mf_head_bias = L.mf_head_bias;
    a.mf_lds_bytes = L.mf_lds_bytes;
    for (int i = 0; i < 16; i++)
    {
      a.mf_layer_tiles[i] = L.mf_layer_tiles[i];
      a.mf_layer_bias[i] = L.mf_layer_bias[i];
    }
    // AUTO: small cells (hidden <= 4) one gate row per lane and four streams per wavefront, cells of 5 .. 32 units two
    // gate rows per lane and one stream per wavefront, else the matrix-core kernel (16 streams per wavefront);
    // NAM_HIP_KERNEL_A1_MFMA forces the matrix-core kernel; NAM_HIP_KERNEL_GENERIC: lanes = streams
    if (b->kernel != NAM_HIP_KERNEL_GENERIC && b->kernel != NAM_HIP_KERNEL_A1_MFMA && lstm_row_eligible(a))
    {
      a.ps = persist_args(b);
      NAM_HIP_CHECK(launch_lstm_row(a, s));
    }
    else if (b->kernel != NAM_HIP_KERNEL_GENERIC && b->kernel != NAM_HIP_KERNEL_A1_MFMA && lstm_wide_eligible(a))
    {
      a.ps = persist_args(b);
      NAM_HIP_CHECK(launch_lstm_wide(a, s));
    }
    else if (L.mf_ok && b->kernel != NAM_HIP_KERNEL_GENERIC)
      NAM_HIP_CHECK(launch_lstm_mfma(a, s));
    else
    {
      // a cell whose columns exceed a CU's LDS keeps them in global memory (the reference has no size limit,
      // lstm.cpp:31-68): slower, but it runs
      const long need = lstm_scratch_floats(a);
      if (need > g.scratch_floats)
      {
        NAM_HIP_CHECK(hipStreamSynchronize(s));
        if (g.d_scratch)
          NAM_HIP_CHECK(hipFree(g.d_scratch));
        g.d_scratch = nullptr;
        g.scratch_floats = 0;
        NAM_HIP_CHECK(hipMalloc(&g.d_scratch, (size_t)need * sizeof(float)));
        g.scratch_floats = need;
      }
      a.scratch = g.d_scratch;
      NAM_HIP_CHECK(launch_lstm(a, s));
    }
  }
  return NAM_HIP_OK;
}

// DSP::prewarm (NAM/dsp.cpp:67-101): process whole max_frames-sized buffers of silence until at
// least prewarm_samples have gone through.
int prewarm_frames(const nam_hip_batch* b, const Plan& p)
{
  const int bs = std::max(b->max_frames, 1);
  if (p.prewarm_samples <= 0)
    return 0;
  return (p.prewarm_samples + bs - 1) / bs * bs;
}

// `first_stream`: one of the n streams (its state seeds the prewarm cache).
int reset_streams(nam_hip_batch* b, WidthGroup& g, const int* d_map, int n, bool prewarm, int first_stream)
{
  if (n <= 0)
    return NAM_HIP_OK;
  const Plan& p = *g.plan;
  const int frames = prewarm ? prewarm_frames(b, p) : 0;
  const bool all = n == (int)g.streams.size();
  if (p.arch == ARCH_WAVENET)
  {
    if (frames > 0 && g.d_prewarm)
    {
      // a state cached by the same kernel over the same number of frames: copy it (every stream's is identical)
      const int kernel = kernel_for_launch(b, g, frames);
      const int fam = state_family_of(p, kernel);
      if (g.prewarm_kernel == kernel && g.prewarm_len == frames && (all || g.state_family < 0 || g.state_family == fam))
      {
        NAM_HIP_CHECK(launch_fill_state(g.d_state, g.state_stride, d_map, n, g.d_prewarm, p.state_floats, p.state_floats, b->stream));
        g.state_family = fam;
        return NAM_HIP_OK;
      }
    }
    NAM_HIP_CHECK(launch_fill_state(g.d_state, g.state_stride, d_map, n, nullptr, 0, p.state_floats, b->stream));
    if (all)
      g.state_family = -1; // every stream of the group is zeroed: either layout may follow
  }
  if (frames > 0)
  {
    const int rc = launch_group(b, g, d_map, n, nullptr, nullptr, frames, 0, b->stream);
    if (rc != NAM_HIP_OK)
      return rc;
    if (p.arch == ARCH_WAVENET && first_stream >= 0)
    {
      if (!g.d_prewarm)
        NAM_HIP_CHECK(hipMalloc(&g.d_prewarm, (size_t)p.state_floats * sizeof(float)));
      NAM_HIP_CHECK(hipMemcpyAsync(g.d_prewarm, g.d_state + (size_t)first_stream * g.state_stride,
                                   (size_t)p.state_floats * sizeof(float), hipMemcpyDeviceToDevice, b->stream));
      g.prewarm_kernel = kernel_for_launch(b, g, frames);
      g.prewarm_len = frames;
    }
  }
  return NAM_HIP_OK;
}

// ---- persistent block mode -------------------------------------------------------------------------------------
constexpr int kGraceUs = 40; // how long a fresh launch looks for the doorbell it was started for
constexpr int kPersistMaxFrames = 2048; // buffers up to this long go through the session as n_frames / 64 commands
// the state layout the session's kernel keeps (WaveNets only)
int persist_family(const nam_hip_batch* b, const WidthGroup& g)
{
  return persist_kind(b) == PERSIST_WN_REG ? 2 : state_family_of(*g.plan, NAM_HIP_KERNEL_A1_IL);
}

// Which kernel a persistent session of this batch would run (PERSIST_NONE: the mode does not apply). Every workgroup
// of the session's launch must be on the chip at once — a workgroup that is waiting for a slot consumes nothing while
// the resident ones keep the ring busy — hence the stream limits.
int persist_kind(const nam_hip_batch* b)
{
  const WidthGroup& g = b->groups[b->model->full_width];
  if (!b->ps.enabled)
    return PERSIST_NONE;
  const int cus = std::max(b->n_cus, 1);
  {
    // nam_wn_reg_kernel serves every width group with one launch: a mixed-width batch is one session. Its workgroups
    // are one wavefront with (the largest group's) LDS image: at most four per CU, and no more than fit its 160 KB
    const WrGroupList gs = wr_groups(const_cast<nam_hip_batch*>(b));
    if (gs.n > 0)
    {
      int lds = 1;
      for (int k = 0; k < gs.n; k++)
        lds = std::max(lds, gs.g[k]->plan->wr.lds_bytes);
      const int per_cu = std::min(4, (160 * 1024) / (lds + 512));
      // (more workgroups than the chip holds at once take turns, as below: each turn moves its streams' LDS images in and
      // out once and consumes every command that is there)
      return b->n_streams <= kPersistTurns * per_cu * cus ? PERSIST_WN_REG : PERSIST_NONE;
    }
  }
  if ((int)g.streams.size() != b->n_streams || g.d_map != nullptr)
    return PERSIST_NONE;
  if (g.plan->arch == ARCH_WAVENET)
  {
    // One workgroup per stream holding most of a CU's LDS: `cus` of them are on the chip at once. More streams than
    // that still make a session — the workgroups never wait for a command, so the resident ones drain the ring and
    // leave, the next ones start behind them and consume the same commands (every workgroup resumes from its own
    // count) — in as many turns as it takes; bounded so that the completion words stay a short scan for the host.
    const int wg_limit = kPersistTurns * cus;
    if (g.plan->a1.valid && g.plan->a1.il_ok && g.plan->a1.p2_ok && b->n_streams <= wg_limit
        && (b->kernel == NAM_HIP_KERNEL_AUTO || b->kernel == NAM_HIP_KERNEL_A1_IL))
      return PERSIST_A1_P2;
    if (!b->no_pipe && g.plan->a1.valid && kq_runs(b, *g.plan) && !g.plan->a1.ws_ok && b->n_streams <= wg_limit
        && pick_kernel(b, g) == NAM_HIP_KERNEL_A1_MFMA)
      return PERSIST_KQ;
    return PERSIST_NONE;
  }
  if (g.plan->arch == ARCH_LSTM && b->kernel == NAM_HIP_KERNEL_AUTO)
  {
    const LSTMPlan& L = g.plan->lstm;
    if (L.hidden >= 1 && L.hidden <= 4 && L.n_layers >= 1 && L.n_layers <= 2 && L.input_size >= 1 && L.input_size <= 2
        && L.in_ch == L.input_size && L.out_ch >= 1 && L.out_ch <= 16 && (b->n_streams + 3) / 4 <= 8 * cus)
      return PERSIST_LSTM_ROW;
    if (L.hidden >= 5 && L.hidden <= 32 && L.n_layers >= 1 && L.n_layers <= 2 && L.input_size >= 1 && L.input_size <= 2
        && L.in_ch == L.input_size && L.out_ch >= 1 && L.out_ch <= 16 && b->n_streams <= 4 * cus) // one wavefront per SIMD
      return PERSIST_LSTM_WIDE;
  }
  return PERSIST_NONE;
}

// (Re)starts the session's launch: every workgroup resumes behind the commands it has consumed so far and runs until
// it finds the ring empty. `grace_us`: how long the launch looks for its first doorbell (rung just before, on the
// caller's hardware queue, so it may land after the launch has started).
int persist_launch(nam_hip_batch* b, int grace_us, long long seq0 = -1, unsigned long long cmd0 = 0)
{
  PersistSession& ps = b->ps;
  WidthGroup& g = b->groups[b->model->full_width];
  if (ps.need_order)
  {
    // the first launch of the session starts behind whatever the batch's own stream still has in flight (a reset, a
    // prewarm, an ordinary launch)
    NAM_HIP_CHECK(hipEventRecord(ps.order, b->stream));
    NAM_HIP_CHECK(hipStreamWaitEvent(ps.kstream, ps.order, 0));
    ps.need_order = false;
  }
  ps.n_launches++;
  ps.grace = grace_us * 100;
  ps.seq0 = seq0;
  ps.cmd0 = cmd0;
  // the workgroups set the top bit of their completion word when they leave: cleared here, "all set" = no launch of
  // the session is running any more (cheaper for the host to look at than hipStreamQuery on a busy stream)
  for (int w = 0; w < ps.n_wg; w++)
    __atomic_and_fetch(&ps.h_words[ps.done_off + w], 0x7fffffffu, __ATOMIC_RELAXED);
  ps.outstanding = true;
  // ticketed host buffers: nam_a1_q_kernel / nam_kq_kernel store the per-buffer completion word (p_cmd_done) behind every
  // command's results (their p_prog, like every kernel's, is ring bookkeeping every 16 commands — never a completion signal;
  // the other kernels' tickets complete when the launch has left)
  {
    const Plan& p = *g.plan;
    // ... and, round 6, a BLOCKING host caller that hands one buffer in after the other (nam_hip_batch::blocking_linger):
    // nam_a1_p4_kernel — what the official topology's short blocking calls run — publishes the word too, so the call waits
    // for its own command and the next call finds the launch still there (no launch, no prologue, no retirement per call)
    (void)p;
    ps.cmd_done_published = (b->pipe_session || b->blocking_linger) && ps.out_is_host && !b->no_pipe
                        && (ps.kind == PERSIST_A1_P2 || ps.kind == PERSIST_KQ); // (pipe_session: never the short-burst rule)
    // ... and linger: a workgroup that finds itself up to date when a launch starts (another one's backlog was the reason for
    // the launch) must not leave at once — the commands to come would find it gone, and the rest of the launch would have to
    // linger and leave before the next launch could pick it up again
    if (ps.cmd_done_published && ps.host_store_ok && ps.n_wg <= b->n_cus)
    {
      ps.grace = std::max(ps.grace, session_linger_ticks(b));
      // "a workgroup of this launch has left" (il_common.h: session_leaving; p_cmd_count[mask + 2] = [kPRing + 1]): none yet
      NAM_HIP_CHECK(hipMemsetAsync(ps.d_cmd_count + kPRing + 1, 0, sizeof(unsigned), ps.kstream));
    }
  }
  const int keep = b->kernel;
  if (ps.kind == PERSIST_A1_P2)
    b->kernel = NAM_HIP_KERNEL_A1_IL;
  b->ps_launching = true;
  namhip::tl_session_stop_event = ps.retired; // (the launch's own completion signal: persist_wait waits on it, not on the stream)
  const int rc = ps.kind == PERSIST_WN_REG
                   ? launch_wr_all(b, wr_groups(b), ps.in_base, ps.out_base, kBlock, ps.stride, ps.kstream)
                   : launch_group(b, g, nullptr, b->n_streams, ps.in_base, ps.out_base, kBlock, ps.stride, ps.kstream);
  namhip::tl_session_stop_event = nullptr;
  b->ps_launching = false;
  b->kernel = keep;
  return rc;
}

// Watchdog of the host's spins on the session's completion words: the resident launch normally answers within
// microseconds, so the spin itself stays a plain memory poll; every 4,096 polls it looks at the launch's stream — a
// launch that has ENDED (or failed: a trap in the kernel, a memory fault, a GPU reset) without every workgroup having
// set its "left" bit will never set it —, at the words the workgroups publish (progress every 16 commands, the count
// when they leave: any change restarts the clock) and at the clock: NAM_HIP_PERSIST_TIMEOUT_MS without ANY workgroup
// moving is a device failure (tests/test_gpu_tickets.py: test_watchdog_*: a launch kept off the CUs by another process).
// Returns NAM_HIP_OK to keep spinning, 1 when the launch is known to have ended (the caller re-reads the words once
// more), or an error.
struct PersistWatch
{
  long polls = 0;
  unsigned long long seen = 0;
  std::chrono::steady_clock::time_point t0{};
  int check(nam_hip_batch* b)
  {
    if ((++polls & 4095) != 0)
      return NAM_HIP_OK;
    const auto now = std::chrono::steady_clock::now();
    unsigned long long sig = 0;
    for (int w = 0; w < 2 * b->ps.done_off; w++)
      sig += __atomic_load_n(&b->ps.h_words[w], __ATOMIC_RELAXED);
    if (polls == 4096 || sig != seen)
      t0 = now;
    seen = sig;
    const hipError_t q = hipStreamQuery(b->ps.kstream);
    if (q == hipSuccess)
      return 1;
    if (q != hipErrorNotReady)
      return fail(NAM_HIP_ERR_DEVICE, std::string("persistent session: the resident launch failed: ") + hipGetErrorString(q));
    if (std::chrono::duration_cast<std::chrono::milliseconds>(now - t0).count() > b->ps.timeout_ms)
      return fail(NAM_HIP_ERR_DEVICE, "persistent session: the resident launch made no progress for "
                                        + std::to_string(b->ps.timeout_ms) + " ms (NAM_HIP_PERSIST_TIMEOUT_MS)");
    return NAM_HIP_OK;
  }
};

// Blocks until every submitted command has been consumed by every workgroup and its results are visible.
// `caller`: the stream the doorbells were rung on.

int persist_wait(nam_hip_batch* b, hipStream_t caller, unsigned target, bool whole);
inline void push_out_host_stores();
int persist_flush(nam_hip_batch* b, hipStream_t caller)
{
  return b->ps.active ? persist_wait(b, caller, b->ps.seq, true) : NAM_HIP_OK;
}

// `whole`: every submitted command (target == seq) and the launch gone. Otherwise: the first `target` commands of the
// session rendered and visible — the launch may run on (the per-buffer completion word p_cmd_done counts then, if the launch publishes it).
int persist_wait(nam_hip_batch* b, hipStream_t caller, unsigned target, bool whole)
{
  PersistSession& ps = b->ps;
  if (!ps.active)
    return NAM_HIP_OK;
  bool told_leave = false;
  if (whole && ps.outstanding && ps.cmd_done_published && ps.host_store_ok && ps.n_wg <= b->n_cus)
  {
    // a lingering launch: tell it that nothing follows command `seq` (kPRingTail)
    __atomic_store_n(&ps.d_ring[kPRing], (unsigned long long)ps.seq, __ATOMIC_RELEASE);
    push_out_host_stores();
    told_leave = true;
  }
  PersistWatch watch;
  bool ended = false; // the stream reported the launch complete: its words are final
  // The workgroups publish their count (behind a release fence behind their last results) when they LEAVE — which
  // they do as soon as they find the ring empty. The host watches those words rather than the launch's completion
  // signal, which takes an interrupt round trip longer.
  bool delivered = false;
  int relaunches = 0;
  for (;;)
  {
    if (!whole && ps.cmd_done_published)
    {
      // ONE word: stored by the last workgroup through the last command of the buffer, behind everybody's results
      // (A1Args::p_cmd_done). A short spin on it between looks at the launch itself (the 2 n_wg words below, which the
      // device writes all the time: a pass over them costs the host microseconds).
      const unsigned* flag = &ps.h_cmd_done[(target - 1u) & (kPRing - 1u)];
      for (int spin = 0; spin < 512; spin++)
      {
        ps.n_polls++;
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == target)
        {
          ps.n_waits++;
          return NAM_HIP_OK; // (whether a launch is still running is the next call's question)
        }
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
      }
    }
    unsigned lo = ~0u, all_left = 0x80000000u;
    for (int w = 0; w < ps.n_wg; w++)
    {
      const unsigned v = __atomic_load_n(&ps.h_words[ps.done_off + w], __ATOMIC_ACQUIRE);
      lo = std::min(lo, v & 0x7fffffffu);
      all_left &= v;
    }
    if (!whole && (int)(lo - target) >= 0)
      return NAM_HIP_OK;
    if (ps.outstanding && !all_left)
    {
      if (ended) // the launch is gone and a workgroup never said goodbye: it died
        return fail(NAM_HIP_ERR_DEVICE, "persistent session: the resident launch ended without every workgroup reporting");
      const int wrc = watch.check(b);
      if (wrc < 0)
        return wrc;
      ended = wrc == 1;
      continue; // the launch is still consuming
    }
    ended = false;
    ps.outstanding = false;
    if ((int)(lo - ps.seq) >= 0)
    {
      ps.flushed = ps.seq;
      ps.flushed_valid = true;
      if (whole && ps.seq != ps.burst_start)
      {
        ps.bursts[2] = ps.bursts[1];
        ps.bursts[1] = ps.bursts[0];
        ps.bursts[0] = ps.seq - ps.burst_start;
        ps.burst_start = ps.seq;
      }
      // every workgroup has published and left; the launch itself retires a moment later (end-of-kernel release). Waiting on the
      // dispatch's own signal costs ~1.4 us and leaves nothing pending on the session's stream: a device-wide synchronize behind
      // this flush (a host that fences per burst: bench.py's timed regions) finds the queue empty instead of pushing a marker
      // through it (~11 us)
      if (whole && ps.retired && ps.n_launches > 0)
        NAM_HIP_CHECK(hipEventSynchronize(ps.retired));
      if (told_leave)
      {
        // the session goes on after a flush: the "leave" word must not stay at this count, or a later launch whose workgroups
        // stand exactly there would leave at once instead of lingering (il_common.h: session_wait_command, `leave == tag - 1`)
        __atomic_store_n(&ps.d_ring[kPRing], ~0ull, __ATOMIC_RELEASE);
        push_out_host_stores();
      }
      return NAM_HIP_OK;
    }
    // no launch running, buffers outstanding: either the commands have not all been delivered yet or a workgroup
    // left just before one landed. Make sure of the former, then run the launch again (it resumes where each stopped).
    if (!delivered)
    {
      NAM_HIP_CHECK(hipStreamSynchronize(caller ? caller : b->stream));
      if (ps.last_caller && ps.last_caller != caller)
        NAM_HIP_CHECK(hipStreamSynchronize(ps.last_caller));
      delivered = true;
    }
    ps.n_flush_relaunches++;
    if (stats_on() && !whole && ps.n_flush_relaunches <= 6)
    {
      unsigned mn = ~0u, mx = 0u;
      int behind = 0;
      for (int w = 0; w < ps.n_wg; w++)
      {
        const unsigned d = ps.h_words[ps.done_off + w] & 0x7fffffffu;
        mn = std::min(mn, d), mx = std::max(mx, d);
        behind += (int)(d - target) < 0 ? 1 : 0;
      }
      std::fprintf(stderr, "nam_hip relaunch from a wait: target %u, submitted %u, workgroups' counts %u .. %u, %d behind the target\n", target, ps.seq, mn, mx, behind);
      if (ps.h_why)
      {
        int hist[2][5] = {{0}};
        long long ex[2] = {0, 0};
        for (int w = 0; w < ps.n_wg; w++)
        {
          const long long y = ps.h_why[w];
          const int grp = (int)((ps.h_words[ps.done_off + w] & 0x7fffffffu) - target) < 0 ? 0 : 1;
          hist[grp][std::min<int>((int)(y >> 56) & 7, 4)]++;
          ex[grp] = y;
          ps.h_why[w] = 0;
        }
        for (int g = 0; g < 2; g++)
          std::fprintf(stderr, "   %s the target: left without a reason recorded %d, leave word %d, cap %d, everybody through %d, somebody left %d; e.g. %s loop, all through %lld, own count %lld\n",
                       g ? "at / beyond" : "behind", hist[g][0], hist[g][1], hist[g][2], hist[g][3], hist[g][4], ((ex[g] >> 48) & 1) ? "start" : "end-of-buffer",
                       (ex[g] >> 24) & 0xffffff, ex[g] & 0xffffff);
      }
    }
    if (++relaunches > 64)
      return fail(NAM_HIP_ERR_DEVICE, "persistent session: submitted buffers were not consumed");
    const int rc = persist_launch(b, 0);
    if (rc != NAM_HIP_OK)
      return rc;
  }
}

int persist_stop(nam_hip_batch* b)
{
  PersistSession& ps = b->ps;
  if (!ps.active)
    return NAM_HIP_OK;
  const int rc = persist_flush(b, ps.last_caller ? ps.last_caller : b->stream);
  // the state is the caller's again only when the launch has gone: a successful whole flush has waited on the launch's own
  // completion signal (persist_wait; 1.4 us — a stream synchronize pushes a marker through the queue, 11 us: profiles/r05/sync_tail.txt)
  if (rc != NAM_HIP_OK || !ps.retired)
    NAM_HIP_CHECK(hipStreamSynchronize(ps.kstream));
  ps.active = false;
  return rc;
}

// Everything a session needs that does not depend on its window — command ring, completion words, the launch's stream and events,
// the low-latency sibling kernel's code object — allocated OUTSIDE the audio path: nam_hip_batch_set_persistent and nam_hip_batch_reset
// call this (the reference's contract: process() never allocates, Reset / get_dsp run on a non-real-time thread; NAM/dsp.h:97,163), so
// the first buffer of a session costs what every first buffer of a launch costs instead of ~7 ms of allocations (256 streams).
void persist_free(nam_hip_batch* b);
static int persist_prepare_alloc(nam_hip_batch* b);
int persist_prepare(nam_hip_batch* b)
{
  if (b->ps.prepared)
    return NAM_HIP_OK;
  // all or nothing: a failure half-way (the ring is there, the stream or an event is not) must not look "prepared" to the next
  // call — it would run a session with a null stream or completion word. Everything allocated so far is released, the mode is
  // off again (nam_hip_batch_set_persistent / nam_hip_batch_reset report the error; a later call may try again)
  const int rc = persist_prepare_alloc(b);
  if (rc != NAM_HIP_OK)
  {
    const std::string why = nam_hip_last_error();
    persist_free(b); // (ps = PersistSession(): enabled = false)
    return fail(rc, why);
  }
  b->ps.prepared = true;
  return NAM_HIP_OK;
}
static int persist_prepare_alloc(nam_hip_batch* b)
{
  PersistSession& ps = b->ps;
    ps.host_store_ok = hipExtMallocWithFlags(reinterpret_cast<void**>(&ps.d_ring), (kPRing + kPRingTail) * sizeof(unsigned long long),
                                             hipDeviceMallocFinegrained) == hipSuccess;
    if (!ps.host_store_ok)
    {
      (void)hipGetLastError();
      NAM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&ps.d_ring), (kPRing + kPRingTail) * sizeof(unsigned long long)));
    }
    NAM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&ps.d_cons), (size_t)b->n_streams * sizeof(unsigned)));
    NAM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&ps.d_cmd_count), (kPRing + kPRingTail) * sizeof(unsigned))); // ([kPRing]: the highest command every workgroup is through)
    NAM_HIP_CHECK(hipMemset(ps.d_cmd_count, 0, (kPRing + kPRingTail) * sizeof(unsigned)));
    NAM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&ps.h_cmd_done), kPRing * sizeof(unsigned), hipHostMallocMapped | hipHostMallocCoherent));
    NAM_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&ps.d_cmd_done), ps.h_cmd_done, 0));
    std::memset(ps.h_cmd_done, 0, kPRing * sizeof(unsigned));
    if (stats_on())
    {
      NAM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&ps.h_why), (size_t)b->n_streams * sizeof(long long), hipHostMallocMapped | hipHostMallocCoherent));
      NAM_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&ps.d_why), ps.h_why, 0));
      std::memset(ps.h_why, 0, (size_t)b->n_streams * sizeof(long long));
    }
    NAM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&ps.h_words), 2 * (size_t)b->n_streams * sizeof(unsigned),
                                hipHostMallocMapped | hipHostMallocCoherent));
    NAM_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&ps.d_words), ps.h_words, 0));
    // A stream of the highest priority has a hardware queue of its own: HIP multiplexes streams of one priority onto
    // a few hardware queues, and a doorbell enqueued behind the session's launch on a shared queue would only be
    // rung after the launch has left.
    int prio_lo = 0, prio_hi = 0;
    NAM_HIP_CHECK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    NAM_HIP_CHECK(hipStreamCreateWithPriority(&ps.kstream, hipStreamNonBlocking, prio_hi));
    NAM_HIP_CHECK(hipEventCreateWithFlags(&ps.order, hipEventDisableTiming));
    NAM_HIP_CHECK(hipEventCreateWithFlags(&ps.retired, hipEventDisableTiming));
    NAM_HIP_CHECK(hipMemset(ps.d_ring, 0, kPRing * sizeof(unsigned long long)));
    NAM_HIP_CHECK(hipMemset(ps.d_ring + kPRing, 0xff, kPRingTail * sizeof(unsigned long long))); // (the "leave" word: no count)
    NAM_HIP_CHECK(hipDeviceSynchronize());
    NAM_HIP_CHECK(hipMemset(ps.d_cons, 0, (size_t)b->n_streams * sizeof(unsigned)));
    std::memset(ps.h_words, 0, 2 * (size_t)b->n_streams * sizeof(unsigned));
    ps.seq = 0; // (sequence numbers run on across sessions — no ring slot needs clearing — until they are rebased, below)
    ps.burst_start = 0;
    if (const char* e = std::getenv("NAM_HIP_PERSIST_REBASE_AT"))
      ps.rebase_at = (unsigned)std::max(1l, std::atol(e));
    if (const char* e = std::getenv("NAM_HIP_PERSIST_TIMEOUT_MS"))
      ps.timeout_ms = std::max(1l, std::atol(e));
  {
    // a session of the headline kernel may start its low-latency sibling later (short_bursts): its code object is loaded now,
    // not at the switch (~1.6 ms on first use) — both output forms, the window is not known yet
    const WidthGroup& g0 = b->groups[b->model->full_width];
    if (g0.plan->arch == ARCH_WAVENET && g0.plan->a1.valid && g0.plan->a1.p2_ok && q_runs(b, *g0.plan))
      for (int oh = 0; oh < 2; oh++)
        NAM_HIP_CHECK(preload_a1_p4_session(g0.plan->a1.p2_c0, g0.plan->a1.p2_c1, g0.plan->a1.arr[0].act, oh != 0));
  }
  return NAM_HIP_OK;
}

int persist_start(nam_hip_batch* b, const float* d_in, float* d_out, long stride)
{
  PersistSession& ps = b->ps;
  const int n = b->n_streams; // (a session holds every stream of the batch)
  {
    const int rc = persist_prepare(b); // (no-op when set_persistent / Reset have done it)
    if (rc != NAM_HIP_OK)
      return rc;
  }
  if (ps.seq >= ps.rebase_at || ps.rebase_pending)
  {
    ps.rebase_pending = false;
    // a session starts flushed (persist_stop: every workgroup at exactly `seq`, the launch gone): renumber from 0. Stale
    // ring slots carry tags near the old count, which a small count never matches; cleared anyway.
    NAM_HIP_CHECK(hipStreamSynchronize(ps.kstream));
    NAM_HIP_CHECK(hipMemset(ps.d_ring, 0, kPRing * sizeof(unsigned long long)));
    NAM_HIP_CHECK(hipMemset(ps.d_ring + kPRing, 0xff, kPRingTail * sizeof(unsigned long long))); // (the "leave" word: no count)
    NAM_HIP_CHECK(hipMemset(ps.d_cons, 0, (size_t)b->n_streams * sizeof(unsigned)));
    NAM_HIP_CHECK(hipMemset(ps.d_cmd_count, 0, (kPRing + kPRingTail) * sizeof(unsigned)));
    NAM_HIP_CHECK(hipDeviceSynchronize());
    std::memset(ps.h_cmd_done, 0, kPRing * sizeof(unsigned)); // (tags of the old numbering)
    for (int w = 0; w < b->n_streams; w++)
    {
      ps.h_words[w] = 0u;
      ps.h_words[b->n_streams + w] = 0x80000000u;
    }
    ps.seq = 0;
    ps.burst_start = 0;
    ps.flushed = 0;
    ps.flushed_valid = true;
    ps.outstanding = false;
  }
  const int kind = persist_kind(b);
  if (kind != ps.kind)
  {
    // another kernel, another workgroup count: every workgroup of the new shape starts behind the commands consumed
    // so far (nothing of the old session is in flight: a session ends with a flush)
    std::vector<unsigned> at((size_t)b->n_streams, ps.seq);
    NAM_HIP_CHECK(hipMemcpy(ps.d_cons, at.data(), at.size() * sizeof(unsigned), hipMemcpyHostToDevice));
    for (int w = 0; w < b->n_streams; w++)
    {
      ps.h_words[w] = ps.seq;
      ps.h_words[b->n_streams + w] = ps.seq | 0x80000000u;
    }
    ps.kind = kind;
    ps.flushed = ps.seq;
    ps.flushed_valid = true;
    ps.outstanding = false;
  }
  ps.in_base = d_in;
  ps.out_base = d_out;
  {
    // where the results go decides how nam_a1_p2 / p4 store them (A1Args::p_out_host)
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, d_out) == hipSuccess)
      ps.out_is_host = at.type == hipMemoryTypeHost;
    else
    {
      (void)hipGetLastError(); // (an address the runtime does not know: treated as device memory)
      ps.out_is_host = false;
    }
  }
  ps.stride = stride;
  ps.done_off = b->n_streams;
  ps.n_wg = kind == PERSIST_LSTM_ROW ? (n + 3) / 4 : n;
  ps.active = true;
  ps.need_order = true;
  ps.n_starts++;
  ps.epoch++;
  return NAM_HIP_OK;
}

// One 64-frame buffer for every stream of the batch through the session. Returns 1 when this call cannot be expressed
// as a command of a session (the caller then launches as usual).
int persist_submit_block(nam_hip_batch* b, const float* d_in, float* d_out, long stride, hipStream_t caller);
// A buffer of any multiple of 64 frames (what hosts send: NAM/dsp.h:97 takes any num_frames <= maxBufferSize; plugins run
// 64 ... 1,024) is that many commands, submitted back to back: the session renders them without a kernel boundary in between.
int persist_submit(nam_hip_batch* b, const float* d_in, float* d_out, int n_frames, long stride, hipStream_t caller)
{
  // (longer calls — an offline render of a whole file — are one resident launch of their own: same kernel, no commands)
  if (n_frames <= 0 || n_frames % kBlock != 0 || n_frames > kPersistMaxFrames)
    return 1;
  {
    // A buffer is never split across sessions: whether this one still fits the session — its sequence numbers below the
    // rebase mark, its LAST command inside the 2 GB window the kernels address — is decided once, here, not command by
    // command (a session that ended between two commands of a buffer restarted with the slot pointer as its base: the next
    // slot then lay below it and forced another restart — correct, and silently slow)
    PersistSession& ps = b->ps;
    if (ps.active)
    {
      const long off_last = (d_in + (n_frames - kBlock)) - ps.in_base;
      const bool past_mark = ps.seq + (unsigned)(n_frames / kBlock) >= ps.rebase_at;
      if (past_mark || off_last > 0x1fff0000l)
      {
        // (the session that starts with this buffer renumbers from 0 even if the count itself has not reached the mark yet:
        // otherwise a buffer of several commands would reach it in mid-buffer and be split after all)
        ps.rebase_pending = ps.rebase_pending || past_mark;
        const int rc = persist_stop(b);
        if (rc != NAM_HIP_OK)
          return rc;
      }
    }
  }
  for (int f = 0; f < n_frames; f += kBlock)
  {
    const int rc = persist_submit_block(b, d_in + f, d_out + f, stride, caller);
    if (rc != NAM_HIP_OK)
      return rc < 0 ? rc : (f == 0 ? rc : fail(NAM_HIP_ERR_DEVICE, "persistent session: a buffer was split across sessions"));
  }
  return NAM_HIP_OK;
}

int persist_submit_block(nam_hip_batch* b, const float* d_in, float* d_out, long stride, hipStream_t caller)
{
  PersistSession& ps = b->ps;
  if (ps.active)
  {
    const long off_in = d_in - ps.in_base, off_out = d_out - ps.out_base;
    // a different window: the session ends, the next one starts here. So does a session whose sequence numbers have reached
    // the rebase mark: one that never ends by itself (the C++ adapter's default: a session per Reset, flushes only) would
    // otherwise run its count into bit 31, the "left" flag of the completion words; persist_start renumbers from 0.
    if (stride != ps.stride || off_in != off_out || off_in < 0 || off_in > 0x1fff0000l /* (the kernels address a window through a 2 GB buffer descriptor) */ || ps.seq >= ps.rebase_at)
    {
      const int rc = persist_stop(b);
      if (rc != NAM_HIP_OK)
        return rc;
    }
  }
  if (!ps.active)
  {
    const int rc = persist_start(b, d_in, d_out, stride);
    if (rc != NAM_HIP_OK)
      return rc;
  }
  // never lap a workgroup by a whole ring (they report their progress every 16 commands and when they leave): the
  // host waits here for the slowest one to move on — back-pressure, at the pace the device consumes
  if ((ps.seq & 63u) == 0u)
  {
    PersistWatch watch;
    for (;;)
    {
      unsigned lo = ~0u, all_left = 0x80000000u;
      for (int w = 0; w < ps.n_wg; w++)
      {
        const unsigned d = __atomic_load_n(&ps.h_words[ps.done_off + w], __ATOMIC_RELAXED);
        lo = std::min(lo, std::max(__atomic_load_n(&ps.h_words[w], __ATOMIC_RELAXED), d & 0x7fffffffu));
        all_left &= d;
      }
      if (ps.seq - lo < kPRing - 128)
        break;
      if (!ps.outstanding || all_left) // nothing is consuming (a launch left early): the flush starts it again
      {
        const int rc = persist_flush(b, caller);
        if (rc != NAM_HIP_OK)
          return rc;
      }
      else
      {
        const int wrc = watch.check(b); // (1 = the launch has ended: the next pass sees every "left" bit and flushes)
        if (wrc < 0)
          return wrc;
      }
    }
  }
  // Is a launch of the session needed? None running (none yet, or the last one found the ring empty and left: every
  // workgroup has set the top bit of its completion word). A launch that is still running picks the command up
  // itself, or leaves just before it lands, in which case the next call (or the flush) starts it again.
  bool idle = !ps.outstanding;
  bool uniform = idle && ps.flushed_valid && ps.flushed == ps.seq; // every workgroup has consumed exactly seq commands
  if (!idle)
  {
    const unsigned left = ps.seq | 0x80000000u;
    idle = uniform = true;
    for (int w = 0; w < ps.n_wg && idle; w++)
    {
      const unsigned v = __atomic_load_n(&ps.h_words[ps.done_off + w], __ATOMIC_ACQUIRE);
      idle = (v & 0x80000000u) != 0;
      uniform = uniform && v == left;
    }
    uniform = uniform && idle;
    if (idle)
      ps.outstanding = false;
  }
  const unsigned long long cmd = ((unsigned long long)(ps.seq + 1) << 32) | (unsigned long long)(unsigned)(d_in - ps.in_base);
  const unsigned slot = ps.seq & (kPRing - 1);
  // Nothing in flight on the caller's stream: nothing to order the command behind, the host stores it itself (no
  // device-side write operation, which costs the host ~4 us and the device a small kernel per buffer).
  if (ps.host_store_ok && (!ps.last_caller || ps.last_caller == caller) && hipStreamQuery(caller) == hipSuccess)
  {
    ps.n_host_doorbells++;
    __atomic_store_n(&ps.d_ring[slot], cmd, __ATOMIC_RELEASE);
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_sfence(); // (the BAR mapping may be write-combining: push the store out now)
#else
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
#endif
    if (idle)
    {
      // (when every workgroup stands at the same count, that count and this command travel with the launch itself)
      const int rc = uniform ? persist_launch(b, 0, (long long)ps.seq, cmd) : persist_launch(b, 0);
      if (rc != NAM_HIP_OK)
        return rc;
    }
  }
  else
  {
    if (ps.last_caller && ps.last_caller != caller)
      NAM_HIP_CHECK(hipStreamSynchronize(ps.last_caller)); // commands of two streams: keep them in order
    // the launch first, the stream-ordered store behind it: the two travel on different hardware queues, and the
    // launch looks for its first command for kGraceUs
    if (idle)
    {
      const int rc = persist_launch(b, kGraceUs);
      if (rc != NAM_HIP_OK)
        return rc;
    }
    ps.n_stream_doorbells++;
    NAM_HIP_CHECK(hipStreamWriteValue64(caller, ps.d_ring + slot, cmd, 0));
  }
  ps.seq++;
  ps.flushed_valid = false;
  ps.last_caller = caller;
  for (auto& g : b->groups)
    if (!g.streams.empty() && g.plan->arch == ARCH_WAVENET)
      g.state_family = persist_family(b, g);
  return NAM_HIP_OK;
}

void persist_free(nam_hip_batch* b)
{
  PersistSession& ps = b->ps;
  if (ps.d_ring)
    (void)hipFree(ps.d_ring);
  if (ps.d_cons)
    (void)hipFree(ps.d_cons);
  if (ps.d_cmd_count)
    (void)hipFree(ps.d_cmd_count);
  if (ps.h_cmd_done)
    (void)hipHostFree(ps.h_cmd_done);
  if (ps.h_why)
    (void)hipHostFree(ps.h_why);
  if (ps.h_words)
    (void)hipHostFree(ps.h_words);
  if (ps.kstream)
    (void)hipStreamDestroy(ps.kstream);
  if (ps.order)
    (void)hipEventDestroy(ps.order);
  if (ps.retired)
    (void)hipEventDestroy(ps.retired);
  ps = PersistSession();
}

void free_group(WidthGroup& g)
{
  if (g.d_blob)
    (void)hipFree(g.d_blob);
  if (g.d_ops)
    (void)hipFree(g.d_ops);
  if (g.d_wr_blob)
    (void)hipFree(g.d_wr_blob);
  if (g.d_a1)
    (void)hipFree(g.d_a1);
  if (g.d_state)
    (void)hipFree(g.d_state);
  if (g.d_init)
    (void)hipFree(g.d_init);
  if (g.d_scratch)
    (void)hipFree(g.d_scratch);
  if (g.d_map)
    (void)hipFree(g.d_map);
  if (g.d_prewarm)
    (void)hipFree(g.d_prewarm);
  g = WidthGroup();
}

int build_model(std::shared_ptr<ModelSpec> spec, nam_hip_model** out)
{
  auto m = std::make_unique<nam_hip_model>();
  m->spec = std::move(spec);
  // layer shapes outside nam_wn_reg_kernel's ahead-of-time tables: collected over every plan of the model (the widths of
  // a slimmable WaveNet, the submodels of a container) and compiled as ONE code object (wr_jit.cpp), so that a batch with
  // mixed widths still runs as one launch
  WrShapeSet jit_shapes;
  WrShapeSet* const js = wr_jit_enabled() ? &jit_shapes : nullptr;
  if (m->spec->arch == ARCH_WAVENET && m->spec->wavenet.slimmable)
  {
    // enumerate the distinct widths: one probe ratio per interval between breakpoints
    std::vector<double> bp = slimmable_breakpoints(m->spec->wavenet);
    std::vector<double> probes;
    double lo = 0.0;
    for (double x : bp)
    {
      probes.push_back(0.5 * (lo + x));
      lo = x;
    }
    probes.push_back(0.5 * (lo + 1.0));
    probes.push_back(1.0);
    for (double r : probes)
    {
      const std::vector<int> ch = channels_for_ratio(m->spec->wavenet, r);
      if (std::find(m->width_channels.begin(), m->width_channels.end(), ch) == m->width_channels.end())
      {
        m->width_channels.push_back(ch);
        m->plans.push_back(build_wavenet_plan(slim_wavenet(m->spec->wavenet, ch), js));
      }
    }
    m->full_width = m->width_for_ratio(1.0);
  }
  else if (m->spec->arch == ARCH_CONTAINER)
  {
    // one plan per submodel (a slimmable submodel stays at its full size: ContainerModel never forwards
    // SetSlimmableSize to its children); a fresh container has the last submodel active (container.cpp:49)
    for (const auto& sm : m->spec->submodels)
    {
      m->plans.push_back(build_plan(*sm, js));
      m->width_channels.push_back({});
    }
    m->full_width = (int)m->plans.size() - 1;
  }
  else
  {
    m->plans.push_back(build_plan(*m->spec, js));
    m->width_channels.push_back({});
    m->full_width = 0;
  }
  bool any_jit = false;
  for (const Plan& p : m->plans)
    any_jit = any_jit || (p.wr.ok && p.wr.jit);
  if (any_jit)
  {
    std::string why;
    const std::string module = wr_jit_build(jit_shapes, why);
    for (size_t i = 0; i < m->plans.size(); i++)
    {
      Plan& p = m->plans[i];
      if (!(p.wr.ok && p.wr.jit))
        continue;
      if (!module.empty())
        p.wr.jit_module = module;
      else
      {
        // no compiler / sources here: plan again without the model's own shapes (run-time-flag instantiations if the
        // model fits them, else the other kernels take it)
        const ModelSpec& sp = m->spec->arch == ARCH_CONTAINER ? *m->spec->submodels[i] : *m->spec;
        Plan again = (sp.arch == ARCH_WAVENET && sp.wavenet.slimmable) ? build_wavenet_plan(slim_wavenet(sp.wavenet, m->width_channels[i]))
                                                                       : build_plan(sp);
        if (!again.wr.ok)
          again.wr.why += " [" + why + "]";
        // loud: the model still runs, but off its compiled shapes (run-time-flag instantiations, or another kernel: 6 - 9 x
        // slower, profiles/r03/defit_table.txt). nam_hip_model_info says so (has_a1_kernel bit 5), the description too.
        again.wr.jit_failed = why.empty() ? "unknown reason" : why;
        std::fprintf(stderr, "libnam_hip: %s: nam_wn_reg_kernel could not be compiled for this model's layer shapes (%s); it runs on %s\n",
                     sp.architecture_name.c_str(), again.wr.jit_failed.c_str(),
                     again.wr.ok ? "the run-time-flag instantiations" : "another kernel");
        p = std::move(again);
      }
    }
  }
  *out = m.release();
  return NAM_HIP_OK;
}

// The blocking entry points inside a persistent session: the kernel reads the buffer from and writes it to HOST-MAPPED
// memory (float32 rows [stream][channel][max_frames]); in_f32 / in_f64 and out_f32 / out_f64: exactly one of each.
// The host-mapped windows of the session's blocking (`slots` = 1: nam_hip_batch::in_bar, h_out_map) or ticketed
// (NAM_HIP_PIPE_SLOTS: pipe_in_bar, pipe_h_out_map) entry points. false: no such memory here (the copying path serves the call).
bool host_windows(nam_hip_batch* b, int slots, float*& in_bar, float*& h_out_map, float*& d_out_map, bool& failed, bool prealloc = false)
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

// A row of audio into the PCIe window. Non-temporal stores: the window is write-combining memory, where glibc's memcpy
// (rep movsb from a few KB up) moves 8 GB/s and 16-byte streaming stores 40 (tools/src/host_window_copy.hip,
// profiles/r04/host_window_copy.txt).
inline void copy_to_window(float* dst, const float* src, size_t n)
{
#if defined(__x86_64__)
  size_t i = 0;
  while (i < n && (reinterpret_cast<uintptr_t>(dst + i) & 15u) != 0)
  {
    dst[i] = src[i];
    i++;
  }
  typedef float v4f __attribute__((vector_size(16)));
  typedef float v4f_u __attribute__((vector_size(16), aligned(4)));
  for (; i + 4 <= n; i += 4)
    __builtin_nontemporal_store(*reinterpret_cast<const v4f_u*>(src + i), reinterpret_cast<v4f*>(dst + i));
  for (; i < n; i++)
    dst[i] = src[i];
#else
  std::memcpy(dst, src, n * sizeof(float));
#endif
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

inline void push_out_host_stores()
{
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_sfence(); // (write-combining stores through the BAR: out before the command that points at them)
#else
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
#endif
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

} // namespace

extern "C" {

const char* nam_hip_last_error(void)
{
  return g_last_error.c_str();
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
