// api_launch.cpp — model handles -> plans -> device blobs; which kernel a launch of a width group runs and its arguments;
// Reset / prewarm (NAM/dsp.cpp:67-140). See api_internal.h.
#include "api_internal.h"

namespace namhip
{
namespace api
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
static int wr_jit_function(const std::string& path, int device, int stages, void** fn, bool dense = false, int* dense_ok = nullptr)
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
    can_split = can_split && w.split_op[3] >= 1 && w.split_op[3] < (int)w.ops.size();
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
    // ONLY when the best one-per-SIMD form leaves SIMDs without a wavefront (config 5: 768 streams on 1,024 SIMDs): where it
    // fills the chip exactly (config 4: 512 streams x 2, 1,024 x 1, 256 x 4) a second wavefront per SIMD only adds hand-overs —
    // measured 3.95 vs 3.87 us (512 streams, four waves per stream dense vs two plain) and 7.59 vs 6.94 (1,024 streams)
    const bool plain_fills = ((long)total * stages) % (4l * cus) == 0;
    const std::string& module0 = groups[0]->plan->wr.jit_module;
    if (!module0.empty() && can_split && b->wr_max_stages >= 2 && !plain_fills)
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
  if (stages == 2) // (the kernels read a two-wave launch's cut from [1]: WrGroup::split_op)
    for (int k = 0; k < n_groups; k++)
      a.g[k].split_op[1] = groups[k]->plan->wr.split_op[3];
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

} // namespace api
} // namespace namhip
