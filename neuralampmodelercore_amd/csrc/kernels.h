// kernels.h — kernel argument blocks and host-side launch wrappers (kernel_*.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "plan.h"

namespace namhip
{

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the CURRENT DEVICE's copy of a function: a process that runs
// batches on several GPUs must raise it once per (function, device). One of these per kernel instantiation; devices
// beyond 63 set the attribute on every launch.
struct DynamicLdsLimit
{
  unsigned long long raised = 0; // bit d: done on device d (launches of one batch come from one host thread at a time;
                                 // a lost update between threads only repeats the call)
  hipError_t ensure(const void* fn, int bytes)
  {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess)
      return e;
    const bool tracked = dev >= 0 && dev < 64;
    if (tracked && ((__atomic_load_n(&raised, __ATOMIC_RELAXED) >> dev) & 1ull))
      return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && tracked)
      __atomic_fetch_or(&raised, 1ull << dev, __ATOMIC_RELAXED);
    return e;
  }
};

// A session's resident launch carries its own completion signal: persist_launch (api_session.cpp) sets this thread's stop event
// around the launch call, and the launch sites of the session-capable kernels go through nam_launch — hipExtLaunchKernelGGL puts
// the event's signal on the dispatch packet itself, so the host's wait for "the launch has retired" (nam_hip_batch_flush,
// synchronize, the end of a session) is a wait on that signal: ~1.4 us behind the last workgroup instead of the ~11 us a
// stream / device synchronize takes to push a marker packet through the queue behind a plain launch
// (tools/src/sync_tail.hip, profiles/r05/sync_tail.txt). nullptr (every other launch): a plain launch.
extern thread_local hipEvent_t tl_session_stop_event;
template <typename F, typename... Args>
inline void nam_launch(F kernel, dim3 grid, dim3 block, unsigned lds_bytes, hipStream_t stream, Args... args)
{
  if (tl_session_stop_event)
    hipExtLaunchKernelGGL(kernel, grid, block, lds_bytes, stream, nullptr, tl_session_stop_event, 0, args...);
  else
    hipLaunchKernelGGL(kernel, grid, block, lds_bytes, stream, args...);
}

// Persistent session of a one-wavefront-per-workgroup kernel (persist_wave.h); ring == nullptr: an ordinary launch
struct PersistArgs
{
  unsigned long long* ring = nullptr; // commands: (seq << 32) | frame offset
  int ring_mask = 0; // ring size - 1
  unsigned* cons = nullptr; // device memory: commands consumed per workgroup (where its next launch resumes)
  unsigned* prog = nullptr; // host-mapped: progress (every 16 commands)
  unsigned* done = nullptr; // host-mapped: consumed count | "left" bit, written when the workgroup leaves
  long long seq0 = -1; // >= 0: every workgroup has consumed exactly this many commands (cons is not read) ...
  unsigned long long cmd0 = 0; // ... and this is the next command (the ring is not read for it)
  int grace = 0; // ticks (100 MHz) a fresh launch looks for its first command before it leaves again
};

// I/O convention for every kernel: planar float32 in HBM,
//   in [stream][in_ch ][io_stride]   out [stream][out_ch][io_stride]
// of which frames [0, n_frames) are processed by this launch (the caller may point `in`/`out` into
// the middle of a longer resident buffer; io_stride is the row pitch in floats).
// `in == nullptr` means silence (prewarm); `out == nullptr` discards the output.
struct GenericArgs
{
  const NamOp* ops;
  const float* blob;
  float* state;
  long state_stride; // floats per stream
  const int* stream_map; // optional: blockIdx -> stream index (mixed-width batches); nullptr = identity
  const float* in;
  float* out;
  long io_stride;
  int n_frames;
  int in_ch, out_ch;
  // weights in LDS: the first blob_floats floats of the blob are copied behind the activation rows (float offset
  // w_lds_off) once per launch and conv weights are read from there (broadcast ds_reads: ~64 cycles, pipelined in
  // order) instead of through scalar loads (~200+ cycles each, not pipelinable across uses); 0 = keep scalar loads
  int w_lds_off, blob_floats;
};

struct A1Args
{
  const A1Plan* plan; // device copy
  const float* blob;
  float* state;
  long state_stride;
  const int* stream_map;
  const float* in;
  float* out;
  long io_stride;
  int n_frames;
  float act_p0; // LeakyReLU slope when the arrays use it
  // MFMA kernel: host-known scalars passed by value so the kernel prologue has no dependent loads
  int n_rings, n_mjobs;
  int tiles_off, consts_off; // blob offsets (floats) of the tile areas / consts table
  float head_scale;
  // wave-specialised kernel
  int r1_off; // blob offset of the first array's rechannel column (16 floats)
  int xt_off, n_xt; // blob offset / count of the extra tiles
  int lds_tiles_b, lds_xt_b, lds_cond_b, lds_bytes; // dynamic LDS layout (bytes)
  int prefetch; // the plan's ws_prefetch (mover prefetch depth the descriptors were built for)
  // interleaved-frame kernel (plan.h: A1Plan::il_*)
  int il_jobs, il_real_jobs, il_depth, il_exch;
  int il_consts_b, il_xt_b, il_tiles_b, il_flag_b, il_lds_bytes;
  int act; // the arrays' activation type when it is uniform (nam_a1_p2_kernel's run-time-dispatch instantiation)
  // persistent session of nam_a1_p2_kernel (nullptr = ordinary launch): command ring in device memory, ring size - 1,
  // commands consumed before this launch, per-workgroup progress / completion words in host-mapped memory
  unsigned long long* p_ring;
  int p_ring_mask;
  unsigned* p_cons; // device memory: commands consumed per workgroup (where its next launch resumes)
  unsigned* p_prog; // host-mapped: progress, stored every 16 commands — the host's ring bookkeeping (back-pressure), NEVER a completion signal
                    // (per-buffer completion: p_cmd_done below; the launch's own: p_done = count | exited bit)
  unsigned* p_done;
  long long p_seq0; // >= 0: every workgroup has consumed exactly this many commands (p_cons is not read) ...
  unsigned long long p_cmd0; // ... and this is the next command (the ring is not read for it)
  int p_grace; // ticks (100 MHz) a fresh launch looks for its first doorbell before it leaves again
  int p_out_host; // the session's output window is HOST memory (nam_a1_p4_kernel: kOutHost — plain result stores, ring
                  // appends written through, one system-scope release fence before the completion word). 2 (nam_a1_q_kernel,
                  // nam_kq_kernel): ticketed host buffers — results written through and p_cmd_done stored after EVERY command,
                  // behind the results: the host takes a buffer's output while the launch runs on (nam_hip_batch_wait_f32)
  int p_linger; // ticks (100 MHz) the launch looks for the NEXT command when it finds the ring empty, before it leaves (nam_a1_q_kernel,
                // nam_kq_kernel; 0 = 1 us, the other kernels' constant): a ticket session's host hands a buffer in every 5 - 15 us
  // ticketed host buffers (p_out_host == 2): p_cmd_count[c & p_ring_mask] (device memory) counts the workgroups that have finished
  // command c and made its results visible; the last one zeroes it and stores c + 1 to p_cmd_done[c & p_ring_mask] (host-mapped):
  // ONE word for the host to poll per buffer, one PCIe write per command
  unsigned* p_cmd_count;
  unsigned* p_cmd_done;
  long long* dbg; // optional: per-job phase timestamps of workgroup 0 (profiling builds / tools only), else nullptr
};

struct LSTMArgs
{
  const float* blob;
  float* state;
  long state_stride;
  const int* stream_map; // optional: launch position -> stream index; nullptr = identity
  const float* in;
  float* out;
  long io_stride;
  int n_frames;
  int n_streams;
  int n_layers, input_size, hidden, in_ch, out_ch, fast;
  int head_w, head_b;
  int layer_w[16];
  int layer_b[16];
  // nam_lstm_mfma_kernel (plan.h: LSTMPlan::mf_*)
  int mf_off, mf_floats, mf_nt, mf_head_tiles, mf_head_bias, mf_lds_bytes;
  int mf_layer_tiles[16];
  int mf_layer_bias[16];
  PersistArgs ps; // nam_lstm_row_kernel / nam_lstm_wide_kernel
  float* scratch = nullptr; // nam_lstm_kernel<true>: global-memory h / c / gate columns (lstm_scratch_floats)
};

// nam_wn_reg_kernel (plan.h: WrPlan). One launch serves up to kWrMaxGroups WIDTH GROUPS — streams of a slimmable model
// (or container) that currently run different sub-models: workgroups [first, next group's first) belong to group g, each
// with its own op list, weights, LDS layout and state ("LDS repacking" per width); the audio windows are shared.
struct WrGroup
{
  const float* blob; // weights, tables and the op list as the kernel's LDS copy holds them (a multiple of 4 floats)
  float* state; // [stream][state_stride]
  const int* stream_map; // optional: position inside the group -> stream index; nullptr = identity
  long state_stride;
  int n_ops, blob_floats;
  int hist_floats; // floats of ring area behind the positions
  int n_slots; // layers (write positions)
  int tab_rows, n_rows, tab_pf, n_pf, tab_ring, tab_ops; // blob float offsets / entry counts of the tables
  int first; // first workgroup of the group
  int split_op[3]; // pipelined launches: the program's cuts at 1/4, 1/2, 3/4 of its weights (two stages: [1]; four: all three)
  int prog; // per-model code objects with the programs compiled in (NAM_WR_PROGRAMS): which of them this group runs
};
struct WrArgs
{
  WrGroup g[kWrMaxGroups];
  int n_groups;
  const float* in; // nullptr = silence (prewarm)
  float* out; // nullptr = discard
  long io_stride;
  int n_frames;
  int in_ch, out_ch;
  PersistArgs ps;
};

// stages = 2 / 4: that many wavefronts per stream (the op program cut at WrGroup::split_op), lds_bytes including the
// (stages - 1) queues of kWrQueueBytes
hipError_t launch_wn_reg(const WrArgs& a, int n_workgroups, int lds_bytes, bool layers, bool runs, bool rt_layers, int stages,
                         hipStream_t stream);
// the same kernel compiled for one model's own layer shapes (wr_jit.cpp); fn = hipFunction_t of the model's code object
hipError_t launch_wn_reg_jit(void* fn, const WrArgs& a, int n_workgroups, int lds_bytes, int stages, hipStream_t stream);
hipError_t launch_generic(const GenericArgs& a, int n_blocks, int lds_bytes, hipStream_t stream);
hipError_t launch_a1(const A1Args& a, int n_blocks, hipStream_t stream);
hipError_t launch_a1_mfma(const A1Args& a, int n_blocks, int act, hipStream_t stream);
hipError_t launch_a1_p2(const A1Args& a, int n_blocks, int c0, int c1, int act, hipStream_t stream);
// nam_a1_p4_kernel: the same models as a pipeline of wave sets decoupled through LDS (kernel_a1_p4.hip)
hipError_t launch_a1_p4(const A1Args& a, int n_blocks, int c0, int c1, int act, hipStream_t stream);
hipError_t preload_a1_p4_session(int c0, int c1, int act, bool out_host); // (kernel_a1_p4.hip)
// nam_a1_q_kernel (kernel_a1_q.hip): the 16 / 8 official topology (aq_table.h) as twelve one-wave stages with LDS-resident
// rings; a.tiles_off = the plan's q weight block (A1Plan::q_w_off), a.consts_off = the FULL-layout tile area (ws_tiles_off)
bool a1_q_takes(int act); // the activations it is compiled for (Fasttanh, Tanh)
hipError_t launch_a1_q(const A1Args& a, int n_blocks, int act, hipStream_t stream);
// nam_kq_kernel (kernel_kq.hip): the A2 topology (kp_table.h) as a pipeline of twelve one-wave stages, one lane per frame, on
// nam_kt_mfma_kernel's stream state; a.tiles_off = the plan's kq weight block (A1Plan::kq_w_off); kq_takes: the activations it
// is compiled for (LeakyReLU with a slope <= 1, ReLU, Tanh, Fasttanh)
bool kq_takes(int act, float act_p0);
hipError_t launch_kq(const A1Args& a, int n_blocks, int act, hipStream_t stream);
hipError_t launch_kt_mfma(const A1Args& a, int n_blocks, int nk, int channels, int lds_aux_floats, int act,
                          hipStream_t stream);
hipError_t launch_lstm(const LSTMArgs& a, hipStream_t stream);
hipError_t launch_lstm_mfma(const LSTMArgs& a, hipStream_t stream);
hipError_t launch_lstm_row(const LSTMArgs& a, hipStream_t stream); // hidden <= 4: one gate row per lane
bool lstm_row_eligible(const LSTMArgs& a);
hipError_t launch_lstm_wide(const LSTMArgs& a, hipStream_t stream); // 5 .. 32 hidden units: two gate rows per lane
bool lstm_wide_eligible(const LSTMArgs& a);
int lstm_lds_bytes(const LSTMArgs& a);
long lstm_scratch_floats(const LSTMArgs& a); // > 0: the lanes = streams kernel keeps its columns in global scratch
hipError_t launch_fill_state(float* state, long state_stride, const int* stream_map, int n_streams, const float* init,
                             int n_init, int state_floats, hipStream_t stream);

} // namespace namhip
