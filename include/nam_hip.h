/*
 * nam_hip.h — C ABI of the MI355X-native NAM inference core (libnam_hip.so).
 *
 * Drop-in boundary for ONE hot path of sdatkinson/NeuralAmpModelerCore: running many independent
 * audio streams through one .nam model (`nam::get_dsp(path)` + `nam::DSP::process(in, out, n)`).
 * The reference exposes that path as a C++ class API, not a C ABI; the entry points below are what
 * a binding for that path needs, each citing the reference interface it replaces
 * (file:line relative to the reference tree). The C++ adapter in cpp/NAM/ re-creates the
 * reference's `nam::DSP` / `nam::get_dsp` signatures on top of this ABI.
 *
 * Conventions
 *   - every function returns NAM_HIP_OK (0) or a negative error code; the message of the last
 *     error on the calling thread is available from nam_hip_last_error(). No C++ exception ever
 *     crosses this boundary (the reference throws at load time: NAM/nam_file.h:11,
 *     NAM/get_dsp.cpp:116-121, NAM/wavenet/model.cpp:671-682, and asserts at run time
 *     NAM/wavenet/model.cpp:824).
 *   - audio is planar: buffer[stream][channel][frame]; the *_f32 / *_f64 entry points take HOST
 *     pointers (they copy to and from the GPU), the *_device entry point takes DEVICE pointers and
 *     an optional hipStream_t and does not synchronise.
 *   - a `nam_hip_model` is immutable host data (parsed file + device plans); a `nam_hip_batch`
 *     owns the GPU memory (weights, per-stream history) for N streams of one model on one device
 *     and must be used from one host thread at a time, like a reference `nam::DSP` instance.
 */
#ifndef NAM_HIP_H
#define NAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
  #define NAM_HIP_API __attribute__((visibility("default")))
#else
  #define NAM_HIP_API
#endif

#define NAM_HIP_OK 0
#define NAM_HIP_ERR_INVALID_ARGUMENT (-1)
#define NAM_HIP_ERR_FILE (-2) /* nam::NamFileValidationError — NAM/nam_file.cpp:9-40 */
#define NAM_HIP_ERR_MODEL (-3) /* std::runtime_error at load: bad version / config / weight count */
#define NAM_HIP_ERR_UNSUPPORTED (-4) /* valid .nam the device path cannot run (e.g. ConvNet / Linear) */
#define NAM_HIP_ERR_DEVICE (-5) /* HIP runtime error */
#define NAM_HIP_ERR_TOO_MANY_FRAMES (-6) /* num_frames > max_frames (assert in NAM/wavenet/model.cpp:824) */

#define NAM_HIP_ARCH_WAVENET 1
#define NAM_HIP_ARCH_LSTM 2
#define NAM_HIP_ARCH_CONTAINER 3 /* SlimmableContainer: submodels selected by nam_hip_batch_set_slimmable_size */

/* kernel selection for nam_hip_batch_set_kernel */
#define NAM_HIP_KERNEL_AUTO 0
#define NAM_HIP_KERNEL_GENERIC 1 /* op-program interpreter (every WaveNet feature) */
#define NAM_HIP_KERNEL_A1 2 /* register-resident VALU kernel for the plain A1 family (1 wave per stream) */
#define NAM_HIP_KERNEL_A1_MFMA 3 /* fp32-MFMA kernels for the A1 family: kernel size 3 everywhere -> wave-specialised kernel
                                    (4 compute + 4 mover waves per stream); a single layer array with other kernel
                                    sizes (A2) -> K-tap kernel (4 waves per stream). Narrower submodels of a container
                                    that neither can run use NAM_HIP_KERNEL_A1 */

#define NAM_HIP_KERNEL_A1_IL 4 /* interleaved-frame fp32-MFMA kernels of the official WaveNet sizes (two arrays of ten layers,
                                  kernel size 3, dilations 1..512; 16/8, 12/8, 8/4 channels): compute wave w owns frames 4j + w,
                                  so dilations 4..32 are DPP row shifts inside the wave and dilations >= 64 the lane's own ring
                                  rows; launches of more than one buffer run the pipelined forms. Any other topology falls
                                  back to NAM_HIP_KERNEL_A1_MFMA */
#define NAM_HIP_KERNEL_WN_REG 5 /* register-resident WaveNet kernel for narrow, feature-rich models (FiLMs, gating, grouped
                                   1x1s, head1x1, a nested condition_dsp — example_models/wavenet_a2_max.nam): one wavefront
                                   per stream, lane = frame, a layer = one unrolled function per instantiated shape; every
                                   conv reaches at most 64 frames back. AUTO picks it when no A1 kernel takes the model;
                                   falls back like AUTO when the model's shapes are not instantiated */

typedef struct nam_hip_model nam_hip_model;
typedef struct nam_hip_batch nam_hip_batch;

/* What nam::DSP's getters report — NAM/dsp.h:100-149,153, NAM/slimmable.h:23-29. */
typedef struct nam_hip_model_info
{
  int32_t architecture; /* NAM_HIP_ARCH_* */
  int32_t in_channels; /* DSP::NumInputChannels  dsp.h:104 */
  int32_t out_channels; /* DSP::NumOutputChannels dsp.h:108 */
  int32_t prewarm_samples; /* DSP::GetPrewarmSamples dsp.h:153 (WaveNet: model.cpp:653-658; LSTM: lstm.cpp:127-134) */
  double expected_sample_rate; /* DSP::GetExpectedSampleRate dsp.h:100; -1.0 when the file has none */
  int32_t has_loudness; /* DSP::HasLoudness dsp.h:137 */
  int32_t has_input_level; /* DSP::HasInputLevel */
  int32_t has_output_level; /* DSP::HasOutputLevel */
  int32_t is_slimmable; /* dynamic_cast<SlimmableModel*> succeeds — slimmable.h:13 */
  double loudness; /* DSP::GetLoudness dsp.h:125 */
  double input_level; /* DSP::GetInputLevel dsp.h:116 */
  double output_level; /* DSP::GetOutputLevel dsp.h:133 */
  int64_t num_weights;
  int32_t fast_tanh; /* load-time switch replacing the global Activation::enable_fast_tanh (activations.cpp:168) */
  int32_t has_a1_kernel; /* bit 0: the A1 VALU kernel can run this model; bit 1: one of the A1 MFMA kernels can; bits 2 and 3
                            (always equal since 0.2.1): the interleaved-frame MFMA kernels can — the official WaveNet sizes, job tables compiled in;
                            bit 4: nam_wn_reg_kernel can; bit 5: the model asked for nam_wn_reg_kernel compiled for its own layer
                            shapes and that compile was not available here (no compiler / sources / private cache directory:
                            a line on stderr and NAM_HIP_FIELD_DESCRIPTION say why) — it runs, on a slower form */
  int64_t state_bytes_per_stream; /* HBM history per stream */
  char version[32]; /* .nam "version" */
} nam_hip_model_info;

/* Message of the last failure on this thread ("" if none). */
NAM_HIP_API const char* nam_hip_last_error(void);

/* ---- loading: nam::get_dsp(path / json) — NAM/get_dsp.h:85-116, NAM/get_dsp.cpp:156-273 ----
 * fast_tanh != 0 mirrors calling nam::activations::Activation::enable_fast_tanh() before get_dsp
 * (tools/benchmodel.cpp:69-73): "Tanh" layers use the rational approximation (activations.h:91-98)
 * and LSTM cells use fast_sigmoid / fast_tanh (lstm.cpp:48-58). */
NAM_HIP_API int nam_hip_model_load(const char* nam_path, int fast_tanh, nam_hip_model** out_model);
NAM_HIP_API int nam_hip_model_load_json(const char* json_text, int fast_tanh, nam_hip_model** out_model);

/* Everything the reference keeps in process-globals around get_dsp, as explicit load options:
 *   fast_tanh  Activation::enable_fast_tanh()                       NAM/activations.cpp:168-177
 *   luts       Activation::enable_lut(function_name, min, max, n)   NAM/activations.cpp:189-212: "Tanh", "Sigmoid" or
 *              "SiLU" layers of the model use FastLUTActivation (NAM/activations.h:371-422: clamp, linear
 *              interpolation in a table of n points built with the host's libm). A table wins over fast_tanh, as it
 *              does when enable_lut is called after enable_fast_tanh. Any other name fails with the reference's message.
 * Exactly one of nam_path / json_text must be non-NULL; options == NULL means all defaults. */
typedef struct nam_hip_lut
{
  const char* function_name;
  float min_x, max_x;
  int32_t n_points;
} nam_hip_lut;
typedef struct nam_hip_load_options
{
  int32_t fast_tanh;
  int32_t n_luts;
  const nam_hip_lut* luts;
  /* != 0: the caller has already run the version gate — verify_config_version with its own registered
   * IVersionSupportChecker objects (NAM/get_dsp.h:19-25,60, get_dsp.cpp:92-128), which can only WIDEN what the core
   * checker accepts — so the library's built-in gate (0.5.0 <= version, minor <= 0.7) is skipped. */
  int32_t version_checked_by_caller;
  /* sizeof(nam_hip_load_options) as the CALLER compiled it (NAM_HIP_LOAD_OPTIONS_INIT sets it). The library reads a field behind
   * the first 16 bytes (fast_tanh, n_luts, luts) only if struct_size says the caller's struct holds it.
   * ABI NOTE (library 0.2.0): this field took the slot of 0.1's `reserved` (that header was 24 bytes: fast_tanh, n_luts, luts,
   * version_checked_by_caller, reserved = 0). A binary built against the 0.1 header therefore passes struct_size = 0 and its
   * version_checked_by_caller is IGNORED — the safe direction: the built-in version gate stays on, a widened file is rejected
   * with the reference's message rather than loaded unchecked. Such callers must be rebuilt against this header
   * (nam_hip_version() reports "0.2.x"); 0 also covers callers that zero-fill, so garbage behind a short struct can never
   * switch the gate off. */
  int32_t struct_size;
} nam_hip_load_options;
#define NAM_HIP_LOAD_OPTIONS_INIT {0, 0, 0, 0, (int32_t)sizeof(nam_hip_load_options)}
/* With version_checked_by_caller the caller's checkers have seen the TOP-LEVEL document only: nested documents (a WaveNet's
 * condition_dsp, a container's submodels) still pass the library's built-in gate, where the reference consults its registry on
 * every get_dsp call (NAM/get_dsp.cpp:92-128). A caller that widens support for nested files must check them itself. */
NAM_HIP_API int nam_hip_model_load_ex(const char* nam_path, const char* json_text, const nam_hip_load_options* options,
                                      nam_hip_model** out_model);
NAM_HIP_API void nam_hip_model_free(nam_hip_model* model);
NAM_HIP_API int nam_hip_model_get_info(const nam_hip_model* model, nam_hip_model_info* info);

/* ---- nam::dspData (NAM/dsp.h:348-357) across the boundary: get_dsp(dspData&) NAM/get_dsp.h:91, get_dsp(path, dspData&
 * returnedConfig) :101, get_dsp(json, dspData& returnedConfig) :109, get_sample_rate_from_nam_file :121 ----
 * nam_hip_model_load_parts builds a model from the fields of a dspData: version, architecture, the "config" value as JSON
 * text, the "metadata" value as JSON text (NULL or "null": none), the weights, expected_sample_rate (-1.0: unknown).
 * It runs verify_config_version + the architecture's config parser + weight binding (get_dsp.cpp:232-261), and fails
 * the way those do. */
NAM_HIP_API int nam_hip_model_load_parts(const char* version, const char* architecture, const char* config_json,
                                         const char* metadata_json, const float* weights, int64_t n_weights,
                                         double expected_sample_rate, const nam_hip_load_options* options,
                                         nam_hip_model** out_model);
/* The dspData of a loaded model (populate_dsp_data, get_dsp.cpp:141-154). String fields: */
#define NAM_HIP_FIELD_VERSION 0
#define NAM_HIP_FIELD_ARCHITECTURE 1
#define NAM_HIP_FIELD_CONFIG_JSON 2 /* the "config" value, compact JSON text */
#define NAM_HIP_FIELD_METADATA_JSON 3 /* the "metadata" value ("null" when the file has none) */
#define NAM_HIP_FIELD_DESCRIPTION 4 /* not dspData: one line about the device plans (which kernels take the model, and why                                        nam_wn_reg_kernel does not when it does not) */
/* Copies up to capacity - 1 bytes + NUL into buf (buf may be NULL); returns the full length (>= 0) or an error code. */
NAM_HIP_API int64_t nam_hip_model_get_string(const nam_hip_model* model, int field, char* buf, int64_t capacity);
/* Copies up to `capacity` weights (out may be NULL); returns their number. A SlimmableContainer has none of its own. */
NAM_HIP_API int64_t nam_hip_model_get_weights(const nam_hip_model* model, float* out, int64_t capacity);
/* get_sample_rate_from_nam_file (NAM/get_dsp.h:121, get_dsp.cpp:275-281) on a .nam document given as a path or as text
 * (exactly one non-NULL): "sample_rate" if present, else -1.0. */
NAM_HIP_API int nam_hip_sample_rate_from_nam(const char* nam_path, const char* json_text, double* out_sample_rate);

/* SlimmableModel::GetSlimmableSizeBreakpoints — NAM/slimmable.h:29, NAM/wavenet/slimmable.cpp:108-121.
 * Writes up to `capacity` values, returns the number available (>= 0) or an error code. */
NAM_HIP_API int nam_hip_model_slimmable_breakpoints(const nam_hip_model* model, double* out, int capacity);

/* ---- batches: N independent streams of one model on one GPU ----
 * Replaces N instances of nam::DSP. `max_frames` plays the role of maxBufferSize in
 * DSP::Reset(sampleRate, maxBufferSize) (NAM/dsp.cpp:130-140): the largest n_frames a process call
 * may pass, and the chunk size used for prewarming (NAM/dsp.cpp:86-100). History starts zeroed
 * (Conv1D::SetMaxBufferSize, NAM/conv1d.cpp:128-149); LSTM state starts from the file's h0/c0
 * (NAM/lstm.cpp:24-28). `device` is the HIP device ordinal. */
NAM_HIP_API int nam_hip_batch_create(const nam_hip_model* model, int device, int n_streams, int max_frames,
                         nam_hip_batch** out_batch);
NAM_HIP_API void nam_hip_batch_destroy(nam_hip_batch* batch);

/* DSP::Reset (NAM/dsp.cpp:130-140): WaveNet history is zeroed; if `prewarm` != 0 the model then
 * processes ceil(prewarm_samples / max_frames) * max_frames frames of silence (DSP::prewarm,
 * NAM/dsp.cpp:67-101). LSTM recurrent state is NOT re-initialised by Reset in the reference
 * (NAM/lstm.cpp has no SetMaxBufferSize override); this call mirrors that: LSTM batches only prewarm. */
NAM_HIP_API int nam_hip_batch_reset(nam_hip_batch* batch, int prewarm);

/* SlimmableModel::SetSlimmableSize(val) for a subset of the streams (NAM/slimmable.h:23,
 * NAM/wavenet/slimmable.cpp:420-431,527-530): the listed streams switch to the sub-model of width
 * ratio_to_channels(ratio) and start from a freshly reset (and, if the batch was last reset with
 * prewarm, prewarmed) state, as the reference's rebuilt model does. stream_ids == NULL selects all. */
NAM_HIP_API int nam_hip_batch_set_slimmable_size(nam_hip_batch* batch, const int* stream_ids, int n_ids, double ratio);

/* DSP::process(NAM_SAMPLE** input, NAM_SAMPLE** output, int num_frames) — NAM/dsp.h:97,
 * NAM/wavenet/model.cpp:822-910, NAM/lstm.cpp:103-125 — for all streams of the batch at once.
 * in  : [n_streams][in_channels ][n_frames]   out : [n_streams][out_channels][n_frames]  (host memory)
 * n_frames must be <= max_frames. Blocks until the output is in `out`. The _f64 form is for callers
 * built with NAM_SAMPLE = double (NAM/dsp.h:18-22); the model itself computes in float32
 * (cast-in NAM/wavenet/model.cpp:817, cast-out :896). */
NAM_HIP_API int nam_hip_batch_process_f32(nam_hip_batch* batch, const float* in, float* out, int n_frames);
NAM_HIP_API int nam_hip_batch_process_f64(nam_hip_batch* batch, const double* in, double* out, int n_frames);

/* Same computation on buffers already resident in HBM (offline re-amping / batch render):
 * d_in / d_out are DEVICE pointers laid out [n_streams][channels][frame_stride]; frames
 * [0, n_frames) of every row are consumed / produced. n_frames may be any length (the kernel walks
 * it in 64-frame blocks, exactly as tools/render.cpp:146-197 walks a file in 64-frame buffers).
 * Enqueues on `hip_stream` (a hipStream_t, NULL = the batch's own stream) and returns without
 * synchronising. */
NAM_HIP_API int nam_hip_batch_process_device(nam_hip_batch* batch, const float* d_in, float* d_out, int n_frames,
                                 int64_t frame_stride, void* hip_stream);

/* Offline render ("re-amp") of whole signals: stream s reads n_frames[s] frames from the planar host buffer
 * in[s] ([in_channels][n_frames[s]]) and writes out[s] ([out_channels][n_frames[s]]). Signals may have different
 * lengths (shorter ones are zero-padded on the device; their tails are discarded). Equivalent to feeding every
 * stream through process() in 64-frame buffers the way tools/render.cpp:129-191 does — for a given kernel the results do
 * not depend on the buffer partition (under NAM_HIP_KERNEL_AUTO a launch of four or more blocks may run another kernel of
 * the same family than a one-block launch: same state, sums associated differently, ~1e-7) — but runs as one resident
 * launch over device-resident audio. Blocking.
 * Replaces the block loop of the reference's render tool (tools/render.cpp:163-197). */
NAM_HIP_API int nam_hip_batch_render_f32(nam_hip_batch* batch, const float* const* in, float* const* out,
                                         const int64_t* n_frames);

/* Persistent block mode (opt-in; nam_a1_q_kernel / nam_a1_p4_kernel, nam_kq_kernel, nam_wn_reg_kernel — mixed slimmable widths included — and
 * the small LSTM kernels): instead of one kernel launch per nam_hip_batch_process_device call, a SESSION launch consumes
 * every call of a multiple of 64 frames (up to 2,048; n_frames / 64 commands) — a command is a 64-bit word in a
 * device-memory ring, stored by the host itself when the call's stream is idle, else by hipStreamWriteValue64 on that
 * stream, so it is ordered behind whatever produced the input there — for as long as the next command is already there
 * when a buffer is finished, and leaves as soon as the ring is empty (a device-wide synchronize never waits for a
 * session; the next call starts the launch again). What a launch per buffer pays on top of the computation (dispatch,
 * cold prologue, state in and out: 4-12 us) is paid once per burst.
 * Consecutive calls must address the same resident window (d_in / d_out of the first call plus a common frame offset,
 * same frame_stride; device memory or host-mapped memory); anything else — another window, a ragged length, Reset,
 * SetSlimmableSize, set_kernel, a render, destroy — ends the session first (the launch writes the streams' state back),
 * transparently. The blocking *_f32 / *_f64 entry points stay inside the session (host-mapped staging).
 * Outputs are NOT ordered on the caller's stream: call nam_hip_batch_flush (or nam_hip_batch_synchronize, or use the
 * blocking *_f32 / *_f64 entry points, which do it) before consuming them.
 * Eligible: one width group on nam_a1_q_kernel / nam_a1_p4_kernel / nam_kq_kernel up to 8 streams per CU (beyond one per CU the workgroups
 * take turns on the chip); every group on nam_wn_reg_kernel up to 8 x min(4, 160 KB / LDS image) streams per CU (in turns too); small LSTMs.
 * Returns 1 if the batch will use the mode, 0 if it is not eligible (the calls then launch as usual).
 * Allocation: this call and nam_hip_batch_reset — the non-real-time side of the reference's contract (NAM/dsp.h:163) — allocate
 * everything a session needs (command ring, completion words, its stream and events, the host windows of the blocking entry points);
 * no process / submit / wait / flush call allocates.
 * First-buffer latency: a session of the official 16 / 8 WaveNet topology whose caller has flushed after at most four buffers three
 * times in a row starts its next launches as nam_a1_p4_kernel (four waves per layer: the first buffer of a launch is through in a
 * few microseconds) instead of nam_a1_q_kernel (sixteen one-wave stages: higher throughput once buffers overlap), and goes back after
 * a longer burst; both work on the same stream state. nam_hip_batch_kernel_name reports what the next launch starts as. */
NAM_HIP_API int nam_hip_batch_set_persistent(nam_hip_batch* batch, int enable);
/* Blocks until every buffer submitted so far has been rendered and is visible (hip_stream: the stream the process
 * calls were issued on; NULL = the batch's own). No-op outside persistent mode. */
NAM_HIP_API int nam_hip_batch_flush(nam_hip_batch* batch, void* hip_stream);

/* DSP::process (NAM/dsp.h:97) for callers that keep buffers IN FLIGHT — a server feeding many streams from a network
 * thread, an offline renderer reading a file ahead: submit copies `in` ([n_streams][in_channels][n_frames], host memory;
 * the caller may reuse it at once), starts the work and returns a ticket; wait blocks until that buffer's output is
 * there and copies it to `out` ([n_streams][out_channels][the submit's n_frames]; NULL: completed and discarded). Up to
 * NAM_HIP_PIPE_SLOTS tickets may be in flight (the next submit fails with NAM_HIP_ERR_INVALID_ARGUMENT until the oldest
 * has been waited for); buffers are rendered in submit order, a ticket is waited for once, in any order.
 * In persistent mode (nam_hip_batch_set_persistent) buffers of a multiple of 64 frames are commands of the session, the
 * input written through the PCIe window, the output stored to host memory by the resident launch itself — nam_a1_q_kernel
 * and nam_kq_kernel publish every command, so a wait returns while the launch runs on with the buffers behind it; other
 * kernels' tickets complete when the launch has drained its ring. (The pipeline of nam_a1_q_kernel holds five to six
 * 64-frame buffers at a time: keep at least eight in flight at that size, four at 256 frames.) Outside persistent mode: pinned staging, copies and the
 * launch enqueued on the batch's stream, an event behind them. Control calls (Reset, SetSlimmableSize, set_kernel,
 * synchronize) complete the tickets in flight first; their outputs stay available to wait. A ticket session's launch
 * looks for the next buffer for 200 us before it leaves (NAM_HIP_TICKET_LINGER_US; it holds its CUs meanwhile). The blocking process calls
 * may be mixed in (tickets in flight complete first). */
#define NAM_HIP_PIPE_SLOTS 16
NAM_HIP_API int nam_hip_batch_submit_f32(nam_hip_batch* batch, const float* in, int n_frames, int64_t* out_ticket);
NAM_HIP_API int nam_hip_batch_wait_f32(nam_hip_batch* batch, int64_t ticket, float* out);
/* The same for callers built with NAM_SAMPLE = double (NAM/dsp.h:18-22): cast in as NAM/wavenet/model.cpp:817, out as :896. */
NAM_HIP_API int nam_hip_batch_submit_f64(nam_hip_batch* batch, const double* in, int n_frames, int64_t* out_ticket);
NAM_HIP_API int nam_hip_batch_wait_f64(nam_hip_batch* batch, int64_t ticket, double* out);

/* Wait for everything enqueued on the batch's own stream (and on the last caller-supplied one). Ends a persistent
 * session: afterwards nothing of the batch is running on the device (a device-wide hipDeviceSynchronize would
 * otherwise wait for the resident launch to expire). */
NAM_HIP_API int nam_hip_batch_synchronize(nam_hip_batch* batch);

/* Choose the kernel (NAM_HIP_KERNEL_*); AUTO picks the fastest kernel the model allows
 * (A1_MFMA > A1 > GENERIC). */
NAM_HIP_API int nam_hip_batch_set_kernel(nam_hip_batch* batch, int kernel);
NAM_HIP_API int nam_hip_batch_get_kernel(const nam_hip_batch* batch);
NAM_HIP_API int nam_hip_batch_n_streams(const nam_hip_batch* batch);
/* Name of the __global__ function the batch's largest stream group currently runs ("nam_a1_mfma_kernel",
 * "nam_kt_mfma_kernel", "nam_a1_kernel", "nam_generic_kernel", "nam_lstm_mfma_reg_kernel", ...): the name
 * rocprofv3 --kernel-trace reports (without template arguments), so measurements can be attributed to the right kernel. */
NAM_HIP_API const char* nam_hip_batch_kernel_name(const nam_hip_batch* batch);
/* The same question for a launch of n_frames (nam_hip_batch_kernel_name answers it for one 64-frame buffer): under
 * NAM_HIP_KERNEL_AUTO a launch that walks four or more blocks — an offline render, the prewarm of Reset — runs the
 * interleaved-frame kernel where a one-block launch runs the wave-specialised one. */
NAM_HIP_API const char* nam_hip_batch_kernel_name_for(const nam_hip_batch* batch, int n_frames);

/* Developer tool, not part of the drop-in surface: runs n_frames of silence through the MFMA kernel's profiling
 * instantiation; out_stamps (96 x 8 int64) receives, per wavefront w of workgroup 0 (row w): barrier cycles, total
 * cycles, and for compute waves five per-job segment sums (tools/mfma_barrier_profile.py). */
NAM_HIP_API int nam_hip_batch_debug_timeline(nam_hip_batch* batch, int n_frames, long long* out_stamps);

/* Number of HIP devices this process can see (what `device` of nam_hip_batch_create indexes; honours
 * HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES). For hosts that spread batches over the GPUs of a node (cpp/NAM/multi_device.h). */
NAM_HIP_API int nam_hip_device_count(int* out_count);

/* The built-in version gate (CoreVersionSupportChecker, NAM/get_dsp.cpp:18-39): 0 = not supported, 1 = partially (newer
 * patch level than the latest fully supported file version), 2 = fully. */
NAM_HIP_API int nam_hip_version_support(const char* nam_file_version);

/* Library identification: "nam_hip <version> gfx950". */
NAM_HIP_API const char* nam_hip_version(void);

#ifdef __cplusplus
}
#endif

#endif /* NAM_HIP_H */
