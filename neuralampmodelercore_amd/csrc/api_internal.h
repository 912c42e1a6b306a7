// api_internal.h — what the translation units behind the C ABI (include/nam_hip.h) share: the model and batch handles, the
// session and ticket records, and the functions they call across files.
//   nam_hip_api.cpp   the extern "C" entry points (argument checks, the order of operations of a call)
//   api_launch.cpp    model -> plans -> device blobs; which kernel a launch runs and its arguments; Reset / prewarm
//   api_session.cpp   persistent block mode: the resident launch, its command ring, completion, the watchdog
//   api_host_io.cpp   host buffers: the windows both sides can reach, blocking calls through a session, tickets
// No exception leaves these files (guarded); errors are codes + nam_hip_last_error.
#pragma once
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
namespace api
{
// the calling thread's last error text (nam_hip_last_error); returns `code`
int fail(int code, const std::string& msg);

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
} // namespace api
} // namespace namhip
using namespace namhip::api;

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

namespace namhip
{
namespace api
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
} // namespace api
} // namespace namhip

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


namespace namhip
{
namespace api
{
// which kernel family a persistent session of a batch runs (api_session.cpp: persist_kind)
enum PersistKind : int
{
  PERSIST_NONE = -1,
  PERSIST_A1_P2 = 0, // nam_a1_q_kernel / nam_a1_p4_kernel (nam_a1_p2_kernel with NAM_HIP_MAX_STAGES=1): one workgroup (most of a CU's LDS) per stream
  PERSIST_WN_REG = 1, // nam_wn_reg_kernel: one wavefront per stream
  PERSIST_LSTM_ROW = 2, // nam_lstm_row_kernel: one wavefront per four streams
  PERSIST_LSTM_WIDE = 3, // nam_lstm_wide_kernel: one wavefront per stream
  PERSIST_KQ = 4 // nam_kq_kernel (the A2 topology): one workgroup (most of a CU's LDS) per stream, as PERSIST_A1_P2
};

constexpr int kPersistTurns = 8; // sessions whose workgroups cannot all be on the chip at once: up to this many turns (they consume the same commands one after the other)
constexpr int kGraceUs = 40; // how long a fresh launch looks for the doorbell it was started for
constexpr int kPersistMaxFrames = 2048; // buffers up to this long go through the session as n_frames / 64 commands

// The non-empty groups of a batch when ALL of them run nam_wn_reg_kernel (then one launch serves the whole batch, and a
// persistent session can too); n = 0 otherwise. (A fixed array: this runs inside process calls, which allocate nothing.)
struct WrGroupList
{
  WidthGroup* g[kWrMaxGroups];
  int n = 0;
};

// api_launch.cpp
int upload_group(nam_hip_batch* b, WidthGroup& g);
int ensure_state(nam_hip_batch* b, WidthGroup& g);
hipError_t quiesce(nam_hip_batch* b);
int state_family_of(const Plan& p, int kernel);
int refresh_map(nam_hip_batch* b, WidthGroup& g);
int pick_kernel(const nam_hip_batch* b, const WidthGroup& g);
const char* group_kernel_name(const nam_hip_batch* b, const WidthGroup& g, int n_frames);
PersistArgs persist_args(const nam_hip_batch* b);
int launch_wr(nam_hip_batch* b, WidthGroup* const* groups, const int* const* maps, const int* counts, int n_groups,
              const float* d_in, float* d_out, int n_frames, long io_stride, hipStream_t s);
WrGroupList wr_groups(nam_hip_batch* b);
int launch_wr_all(nam_hip_batch* b, const WrGroupList& gs, const float* d_in, float* d_out, int n_frames, long io_stride,
                  hipStream_t s);
int kernel_for_launch(const nam_hip_batch* b, const WidthGroup& g, int n_frames);
int launch_group(nam_hip_batch* b, WidthGroup& g, const int* d_map, int n, const float* d_in, float* d_out,
                 int n_frames, long io_stride, hipStream_t s);
int prewarm_frames(const nam_hip_batch* b, const Plan& p);
int reset_streams(nam_hip_batch* b, WidthGroup& g, const int* d_map, int n, bool prewarm, int first_stream);
// api_session.cpp
int persist_family(const nam_hip_batch* b, const WidthGroup& g);
int persist_kind(const nam_hip_batch* b);
int persist_flush(nam_hip_batch* b, hipStream_t caller);
int persist_wait(nam_hip_batch* b, hipStream_t caller, unsigned target, bool whole);
int persist_stop(nam_hip_batch* b);
int persist_prepare(nam_hip_batch* b);
int persist_start(nam_hip_batch* b, const float* d_in, float* d_out, long stride);
int persist_submit(nam_hip_batch* b, const float* d_in, float* d_out, int n_frames, long stride, hipStream_t caller);
int persist_submit_block(nam_hip_batch* b, const float* d_in, float* d_out, long stride, hipStream_t caller);
void persist_free(nam_hip_batch* b);
// api_launch.cpp
void free_group(WidthGroup& g);
int build_model(std::shared_ptr<ModelSpec> spec, nam_hip_model** out);
// api_host_io.cpp
bool host_windows(nam_hip_batch* b, int slots, float*& in_bar, float*& h_out_map, float*& d_out_map, bool& failed, bool prealloc = false);
bool host_mapped_applies(nam_hip_batch* b, int n_frames);
int process_host_mapped(nam_hip_batch* b, const float* in_f32, const double* in_f64, float* out_f32, double* out_f64, int n_frames);
int pipe_submit(nam_hip_batch* b, const float* in, int n_frames, PipeSlot& sl, int slot);
int pipe_wait(nam_hip_batch* b, PipeSlot& sl, int slot, float* out);

inline bool persist_eligible(const nam_hip_batch* b)
{
  return persist_kind(b) != PERSIST_NONE;
}

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

inline double stat_now_us()
{
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

inline bool stats_on()
{
  static const bool on = [] { const char* e = std::getenv("NAM_HIP_SESSION_STATS"); return e && e[0] == '1'; }();
  return on;
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

inline void push_out_host_stores()
{
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_sfence(); // (write-combining stores through the BAR: out before the command that points at them)
#else
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
#endif
}

} // namespace api
} // namespace namhip
