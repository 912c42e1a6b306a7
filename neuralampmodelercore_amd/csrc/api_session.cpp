// api_session.cpp — persistent block mode (nam_hip_batch_set_persistent): one resident launch per session fed through a
// command ring, completion words, flush / stop, the no-progress watchdog. See api_internal.h.
#include "api_internal.h"

namespace namhip
{
namespace api
{

// ---- persistent block mode -------------------------------------------------------------------------------------
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
static int persist_launch(nam_hip_batch* b, int grace_us, long long seq0 = -1, unsigned long long cmd0 = 0)
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

static int persist_prepare_alloc(nam_hip_batch* b);
// Everything a session needs that does not depend on its window — command ring, completion words, the launch's stream and events,
// the low-latency sibling kernel's code object — allocated OUTSIDE the audio path: nam_hip_batch_set_persistent and nam_hip_batch_reset
// call this (the reference's contract: process() never allocates, Reset / get_dsp run on a non-real-time thread; NAM/dsp.h:97,163), so
// the first buffer of a session costs what every first buffer of a launch costs instead of ~7 ms of allocations (256 streams).
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

} // namespace api
} // namespace namhip
