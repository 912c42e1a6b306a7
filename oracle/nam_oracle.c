/*
 * nam_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C float32 restatement of the NeuralAmpModelerCore hot path
 * (WaveNet + LSTM `DSP::process`), written from the reference's behaviour so
 * that the HIP product path has something independent to be checked against.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product (neuralampmodelercore_amd/) never links or calls it.
 *
 * Parity pinning (PINNED): (1) the reference's OWN sources — NAM/*.cpp compiled where they lie by oracle/Makefile.ref
 * against a self-written scalar Eigen stand-in (oracle/eigen_shim/Eigen/Dense; the Eigen submodule is empty here) into
 * oracle/_ref/libnam_ref.so, a2_fast.cpp included — agree with this restatement BIT FOR BIT on every fixture model and on
 * seeded feature-rich models (tests/test_reference_build.py); caveat: that build sums in the stand-in's k-order, not in
 * real Eigen's blocked order (differences below the 5e-5 bar). (2) the reference's primitive-level known-answer tests
 * (tools/test/test_conv1d.cpp, test_conv_1x1.cpp, test_wavenet/test_layer.cpp, test_film.cpp,
 * test_gating_activations.cpp, test_blending_detailed.cpp ...) re-expressed in tests/test_oracle_kat.py. (3) an
 * independent PyTorch-CPU F.conv1d implementation (tests/test_oracle_torch_crosscheck.py). Whole-model golden vectors
 * do not exist in the reference (its model-level tests only assert isfinite): tests/golden/outputs.npz are this
 * oracle's outputs, which (1) ties to the reference's code.
 *
 * Data layout mirrors the reference: matrices are column-major
 * (rows = channels, cols = frames): element (c, f) lives at data[f * C + c].
 * All arithmetic is float32; build with -ffp-contract=off so that every
 * multiply and add is individually rounded (no FMA contraction).
 *
 * Each function cites the reference file:line (relative to /root/reference) it
 * follows.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* Activations — NAM/activations.h:59-133 (scalar fns), :182-369 (classes)    */
/* ------------------------------------------------------------------------- */
enum
{
  ORC_ACT_IDENTITY = 0,
  ORC_ACT_TANH = 1,
  ORC_ACT_HARDTANH = 2,
  ORC_ACT_FASTTANH = 3,
  ORC_ACT_RELU = 4,
  ORC_ACT_LEAKYRELU = 5,
  ORC_ACT_PRELU = 6,
  ORC_ACT_SIGMOID = 7,
  ORC_ACT_SILU = 8,
  ORC_ACT_HARDSWISH = 9,
  ORC_ACT_LEAKYHARDTANH = 10,
  ORC_ACT_SOFTSIGN = 11,
  ORC_ACT_LUT = 13 /* FastLUTActivation, activations.h:371-422 (enable_lut, activations.cpp:189-212) */
};

#define ORC_MAX_SLOPES 64

typedef struct
{
  int type;
  float p[4]; /* LeakyReLU: p[0]=slope; LeakyHardtanh: min_val,max_val,min_slope,max_slope */
  int n_slopes; /* PReLU */
  float slopes[ORC_MAX_SLOPES];
  /* ORC_ACT_LUT: p[0] = min_x, p[1] = max_x, p[2] = base function (TANH / SIGMOID / SILU), p[3] = n points */
  float lut_step, lut_inv_step;
  int lut_n;
  float* lut; /* owned by the process (test infrastructure: never freed) */
} orc_act;

/* activations.h:91-98 */
static inline float orc_fast_tanh(const float x)
{
  const float ax = fabsf(x);
  const float x2 = x * x;
  return (x * (2.45550750702956f + 2.45550750702956f * ax + (0.893229853513558f + 0.821226666969744f * ax) * x2)
          / (2.44506634652299f + (2.44506634652299f + x2) * fabsf(x + 0.814642734961073f * x * ax)));
}
/* activations.h:100-103 */
static inline float orc_fast_sigmoid(const float x)
{
  return 0.5f * (orc_fast_tanh(x * 0.5f) + 1.0f);
}
/* activations.h:64-67 */
static inline float orc_sigmoid(float x)
{
  return 1.0f / (1.0f + expf(-x));
}
/* activations.h:105-108 */
static inline float orc_leaky_relu(float x, float ns)
{
  return x > 0.0f ? x : ns * x;
}

/* FastLUTActivation::lookup, activations.h:391-409 */
static inline float orc_lut_lookup(const orc_act* a, float x)
{
  const float min_x = a->p[0], max_x = a->p[1];
  x = x < min_x ? min_x : (x > max_x ? max_x : x);
  const float f_idx = (x - min_x) * a->lut_inv_step;
  const size_t i = (size_t)f_idx;
  if (i >= (size_t)a->lut_n - 1)
    return a->lut[a->lut_n - 1];
  const float frac = f_idx - (float)i;
  return a->lut[i] + (a->lut[i + 1] - a->lut[i]) * frac;
}

static inline float orc_act_scalar(const orc_act* a, float x, float slope)
{
  switch (a->type)
  {
    case ORC_ACT_LUT: return orc_lut_lookup(a, x);
    case ORC_ACT_IDENTITY: return x;
    case ORC_ACT_TANH: return tanhf(x); /* std::tanh(float) activations.h:188 */
    case ORC_ACT_HARDTANH: /* activations.h:69-73 */
    {
      const float t = x < -1 ? -1 : x;
      return t > 1 ? 1 : t;
    }
    case ORC_ACT_FASTTANH: return orc_fast_tanh(x);
    case ORC_ACT_RELU: return x > 0.0f ? x : 0.0f; /* activations.h:59-62 */
    case ORC_ACT_LEAKYRELU: return orc_leaky_relu(x, a->p[0]);
    case ORC_ACT_PRELU: return orc_leaky_relu(x, slope);
    case ORC_ACT_SIGMOID: return orc_sigmoid(x);
    case ORC_ACT_SILU: return x * orc_sigmoid(x); /* activations.h:115-118 */
    case ORC_ACT_HARDSWISH: /* activations.h:120-128 */
    {
      const float t = x + 3.0f;
      const float clamped = t < 0.0f ? 0.0f : (t > 6.0f ? 6.0f : t);
      return x * clamped * (1.0f / 6.0f);
    }
    case ORC_ACT_LEAKYHARDTANH: /* activations.h:75-89 */
    {
      const float min_val = a->p[0], max_val = a->p[1], min_slope = a->p[2], max_slope = a->p[3];
      if (x < min_val)
        return (x - min_val) * min_slope + min_val;
      else if (x > max_val)
        return (x - max_val) * max_slope + max_val;
      return x;
    }
    case ORC_ACT_SOFTSIGN: return x / (1.0f + fabsf(x)); /* activations.h:130-133 */
  }
  return x;
}

/* Activation::apply(float* data, long size) — flat; PReLU indexes pos % n_slopes
 * on column-major data (activations.h:283-297). */
static void orc_act_apply_flat(const orc_act* a, float* data, long size)
{
  if (a->type == ORC_ACT_IDENTITY)
    return;
  for (long pos = 0; pos < size; pos++)
  {
    const float slope = (a->type == ORC_ACT_PRELU) ? a->slopes[pos % a->n_slopes] : 0.0f;
    data[pos] = orc_act_scalar(a, data[pos], slope);
  }
}

/* Activation::apply(MatrixXf&) on a (rows x 1) column, as used by the gating /
 * blending activations: PReLU uses slopes[row] (activations.h:299-322). */
static void orc_act_apply_column(const orc_act* a, float* col, int rows)
{
  if (a->type == ORC_ACT_IDENTITY)
    return;
  for (int r = 0; r < rows; r++)
  {
    const float slope = (a->type == ORC_ACT_PRELU) ? a->slopes[r % a->n_slopes] : 0.0f;
    col[r] = orc_act_scalar(a, col[r], slope);
  }
}

/* ------------------------------------------------------------------------- */
/* Allocation tracking (one arena list per model)                              */
/* ------------------------------------------------------------------------- */
typedef struct orc_alloc_node
{
  struct orc_alloc_node* next;
} orc_alloc_node;

typedef struct
{
  orc_alloc_node* head;
} orc_arena;

static void* orc_alloc(orc_arena* ar, size_t bytes)
{
  orc_alloc_node* n = (orc_alloc_node*)calloc(1, sizeof(orc_alloc_node) + 16 + bytes);
  if (!n)
  {
    fprintf(stderr, "nam_oracle: out of memory\n");
    abort();
  }
  n->next = ar->head;
  ar->head = n;
  return (void*)((char*)n + sizeof(orc_alloc_node) + (16 - sizeof(orc_alloc_node) % 16) % 16);
}

static void orc_arena_free(orc_arena* ar)
{
  orc_alloc_node* n = ar->head;
  while (n)
  {
    orc_alloc_node* nx = n->next;
    free(n);
    n = nx;
  }
  ar->head = NULL;
}

/* ------------------------------------------------------------------------- */
/* Y[:, f] (op)= Wt^T X[:, f] for a block of frames: y[o][f] = sum_i wt[i][o] x[i][f], */
/* each output summed in input order i = 0..in-1 (the rounding of a plain dot        */
/* product). Frames are handled four at a time and the fixed-N variants exist only   */
/* so the compiler keeps the accumulators in vector registers: every variant does    */
/* exactly the same floating-point operations in the same order per output element.  */
/* accumulate != 0 adds the finished product to Y (per-tap "out += W[k] * in").       */
/* ------------------------------------------------------------------------- */
#define ORC_GEMM_FIXED(N)                                                                                   \
  static void orc_gemm_##N(const float* restrict wt, const float* restrict x, int x_stride, int in_ch,       \
                           int n_frames, float* restrict y, int accumulate)                                 \
  {                                                                                                         \
    int f = 0;                                                                                              \
    for (; f + 4 <= n_frames; f += 4)                                                                       \
    {                                                                                                       \
      float a0[N], a1[N], a2[N], a3[N];                                                                     \
      for (int o = 0; o < N; o++)                                                                           \
        a0[o] = a1[o] = a2[o] = a3[o] = 0.0f;                                                               \
      const float* x0 = x + (size_t)f * x_stride;                                                           \
      const float* x1 = x0 + x_stride;                                                                      \
      const float* x2 = x1 + x_stride;                                                                      \
      const float* x3 = x2 + x_stride;                                                                      \
      for (int i = 0; i < in_ch; i++)                                                                       \
      {                                                                                                     \
        const float* restrict wc = wt + (size_t)i * N;                                                      \
        const float v0 = x0[i], v1 = x1[i], v2 = x2[i], v3 = x3[i];                                         \
        for (int o = 0; o < N; o++)                                                                         \
        {                                                                                                   \
          a0[o] += wc[o] * v0;                                                                              \
          a1[o] += wc[o] * v1;                                                                              \
          a2[o] += wc[o] * v2;                                                                              \
          a3[o] += wc[o] * v3;                                                                              \
        }                                                                                                   \
      }                                                                                                     \
      float* y0 = y + (size_t)f * N;                                                                        \
      if (accumulate)                                                                                       \
        for (int o = 0; o < N; o++)                                                                         \
        {                                                                                                   \
          y0[o] += a0[o];                                                                                   \
          y0[N + o] += a1[o];                                                                               \
          y0[2 * N + o] += a2[o];                                                                           \
          y0[3 * N + o] += a3[o];                                                                           \
        }                                                                                                   \
      else                                                                                                  \
        for (int o = 0; o < N; o++)                                                                         \
        {                                                                                                   \
          y0[o] = a0[o];                                                                                    \
          y0[N + o] = a1[o];                                                                                \
          y0[2 * N + o] = a2[o];                                                                            \
          y0[3 * N + o] = a3[o];                                                                            \
        }                                                                                                   \
    }                                                                                                       \
    for (; f < n_frames; f++)                                                                               \
    {                                                                                                       \
      float a0[N];                                                                                          \
      for (int o = 0; o < N; o++)                                                                           \
        a0[o] = 0.0f;                                                                                       \
      const float* x0 = x + (size_t)f * x_stride;                                                           \
      for (int i = 0; i < in_ch; i++)                                                                       \
      {                                                                                                     \
        const float* restrict wc = wt + (size_t)i * N;                                                      \
        const float v0 = x0[i];                                                                             \
        for (int o = 0; o < N; o++)                                                                         \
          a0[o] += wc[o] * v0;                                                                              \
      }                                                                                                     \
      float* y0 = y + (size_t)f * N;                                                                        \
      for (int o = 0; o < N; o++)                                                                           \
        y0[o] = accumulate ? y0[o] + a0[o] : a0[o];                                                         \
    }                                                                                                       \
  }
ORC_GEMM_FIXED(1)
ORC_GEMM_FIXED(2)
ORC_GEMM_FIXED(3)
ORC_GEMM_FIXED(4)
ORC_GEMM_FIXED(6)
ORC_GEMM_FIXED(8)
ORC_GEMM_FIXED(12)
ORC_GEMM_FIXED(16)

static void orc_gemm(const float* restrict wt, const float* restrict x, int x_stride, int in_ch, int oc, int n_frames,
                     float* restrict y, int accumulate)
{
  switch (oc)
  {
    case 1: orc_gemm_1(wt, x, x_stride, in_ch, n_frames, y, accumulate); return;
    case 2: orc_gemm_2(wt, x, x_stride, in_ch, n_frames, y, accumulate); return;
    case 3: orc_gemm_3(wt, x, x_stride, in_ch, n_frames, y, accumulate); return;
    case 4: orc_gemm_4(wt, x, x_stride, in_ch, n_frames, y, accumulate); return;
    case 6: orc_gemm_6(wt, x, x_stride, in_ch, n_frames, y, accumulate); return;
    case 8: orc_gemm_8(wt, x, x_stride, in_ch, n_frames, y, accumulate); return;
    case 12: orc_gemm_12(wt, x, x_stride, in_ch, n_frames, y, accumulate); return;
    case 16: orc_gemm_16(wt, x, x_stride, in_ch, n_frames, y, accumulate); return;
    default: break;
  }
  for (int f = 0; f < n_frames; f++)
  {
    const float* x0 = x + (size_t)f * x_stride;
    float* y0 = y + (size_t)f * oc;
    for (int o = 0; o < oc; o++)
    {
      float sum = 0.0f;
      for (int i = 0; i < in_ch; i++)
        sum += wt[(size_t)i * oc + o] * x0[i];
      y0[o] = accumulate ? y0[o] + sum : sum;
    }
  }
}

/* ------------------------------------------------------------------------- */
/* Conv1x1 — NAM/dsp.cpp:311-355 (ctor), :363-398 (set_weights_), :436-836     */
/* ------------------------------------------------------------------------- */
typedef struct
{
  int in_ch, out_ch, groups, has_bias;
  float* w; /* dense [out][in] row-major; block-diagonal for grouped */
  float* wt; /* the same matrix stored [in][out] (so the per-frame update vectorises over outputs) */
  float* bias;
  float* out; /* [out_ch x max_buf] column-major */
} orc_conv1x1;

static void orc_conv1x1_init(orc_arena* ar, orc_conv1x1* c, int in_ch, int out_ch, int has_bias, int groups)
{
  c->in_ch = in_ch;
  c->out_ch = out_ch;
  c->groups = groups;
  c->has_bias = has_bias;
  c->w = (float*)orc_alloc(ar, sizeof(float) * (size_t)in_ch * out_ch);
  c->wt = (float*)orc_alloc(ar, sizeof(float) * (size_t)in_ch * out_ch);
  c->bias = has_bias ? (float*)orc_alloc(ar, sizeof(float) * out_ch) : NULL;
  c->out = NULL;
}

static long orc_conv1x1_num_weights(const orc_conv1x1* c)
{
  return (long)c->in_ch * c->out_ch / c->groups + (c->has_bias ? c->out_ch : 0);
}

/* dsp.cpp:363-398: for g, for i in out/g, for j in in/g: W(g*opg+i, g*ipg+j); then bias.
 * The depthwise special case (groups==in==out) consumes the same count in the same order. */
static const float* orc_conv1x1_set_weights(orc_conv1x1* c, const float* w)
{
  const int opg = c->out_ch / c->groups, ipg = c->in_ch / c->groups;
  memset(c->w, 0, sizeof(float) * (size_t)c->in_ch * c->out_ch);
  for (int g = 0; g < c->groups; g++)
    for (int i = 0; i < opg; i++)
      for (int j = 0; j < ipg; j++)
        c->w[(size_t)(g * opg + i) * c->in_ch + (g * ipg + j)] = *(w++);
  for (int o = 0; o < c->out_ch; o++)
    for (int i = 0; i < c->in_ch; i++)
      c->wt[(size_t)i * c->out_ch + o] = c->w[(size_t)o * c->in_ch + i];
  if (c->has_bias)
    for (int i = 0; i < c->out_ch; i++)
      c->bias[i] = *(w++);
  return w;
}

static void orc_conv1x1_set_max_buffer(orc_arena* ar, orc_conv1x1* c, int max_buf)
{
  c->out = (float*)orc_alloc(ar, sizeof(float) * (size_t)c->out_ch * max_buf);
}

/* dsp.cpp:436-449,770-836 (Eigen path): out = W * in (+ bias column-wise).
 * `in` is column-major with column stride in_stride (>= in_ch). */
static void orc_conv1x1_process(orc_conv1x1* c, const float* in, int in_stride, int num_frames)
{
  orc_gemm(c->wt, in, in_stride, c->in_ch, c->out_ch, num_frames, c->out, 0);
  if (c->has_bias)
    for (int f = 0; f < num_frames; f++)
    {
      float* y = c->out + (size_t)f * c->out_ch;
      for (int o = 0; o < c->out_ch; o++)
        y[o] += c->bias[o];
    }
}

/* ------------------------------------------------------------------------- */
/* RingBuffer — NAM/ring_buffer.cpp:7-109                                      */
/* ------------------------------------------------------------------------- */
typedef struct
{
  int channels;
  long cols; /* 2*max_lookback + max_buf */
  long max_lookback;
  long write_pos;
  int max_buf;
  float* storage; /* [channels x cols] column-major */
} orc_ring;

/* ring_buffer.cpp:7-27 */
static void orc_ring_reset(orc_arena* ar, orc_ring* r, int channels, long max_lookback, int max_buf)
{
  r->channels = channels;
  r->max_lookback = max_lookback;
  r->max_buf = max_buf;
  r->cols = 2 * max_lookback + max_buf;
  r->storage = (float*)orc_alloc(ar, sizeof(float) * (size_t)channels * r->cols); /* zeroed */
  r->write_pos = max_lookback;
}

/* ring_buffer.cpp:83-109 */
static void orc_ring_rewind(orc_ring* r)
{
  if (r->max_lookback == 0)
  {
    r->write_pos = 0;
    return;
  }
  const long copy_start = r->write_pos - r->max_lookback;
  memmove(r->storage, r->storage + (size_t)copy_start * r->channels,
          sizeof(float) * (size_t)r->max_lookback * r->channels);
  r->write_pos = r->max_lookback;
}

/* ring_buffer.cpp:29-42 */
static void orc_ring_write(orc_ring* r, const float* in, int in_stride, int num_frames)
{
  if (r->write_pos + num_frames > r->cols)
    orc_ring_rewind(r);
  if (in_stride == r->channels)
    memcpy(r->storage + (size_t)r->write_pos * r->channels, in, sizeof(float) * (size_t)num_frames * r->channels);
  else
    for (int f = 0; f < num_frames; f++)
      memcpy(r->storage + (size_t)(r->write_pos + f) * r->channels, in + (size_t)f * in_stride,
             sizeof(float) * r->channels);
}

/* ring_buffer.cpp:44-57 */
static const float* orc_ring_read(const orc_ring* r, long lookback)
{
  return r->storage + (size_t)(r->write_pos - lookback) * r->channels;
}

/* ------------------------------------------------------------------------- */
/* Conv1D — NAM/conv1d.cpp:11-56 (weights), :128-161 (sizing, prewarm cache),  */
/*          :163-183,666-685,768-775 (Process, generic Eigen path)             */
/* ------------------------------------------------------------------------- */
typedef struct
{
  int in_ch, out_ch, K, dilation, groups, has_bias;
  float* w; /* dense [K][out][in] */
  float* wt; /* dense [K][in][out] */
  float* bias;
  orc_ring ring;
  float* out; /* [out_ch x max_buf] */
  float* cached_col; /* in_ch */
  int has_cached;
} orc_conv1d;

static void orc_conv1d_init(orc_arena* ar, orc_conv1d* c, int in_ch, int out_ch, int K, int has_bias, int dilation,
                            int groups)
{
  memset(c, 0, sizeof(*c));
  c->in_ch = in_ch;
  c->out_ch = out_ch;
  c->K = K;
  c->dilation = dilation;
  c->groups = groups;
  c->has_bias = has_bias;
  c->w = (float*)orc_alloc(ar, sizeof(float) * (size_t)K * in_ch * out_ch);
  c->wt = (float*)orc_alloc(ar, sizeof(float) * (size_t)K * in_ch * out_ch);
  c->bias = has_bias ? (float*)orc_alloc(ar, sizeof(float) * out_ch) : NULL;
  c->cached_col = (float*)orc_alloc(ar, sizeof(float) * in_ch);
}

static long orc_conv1d_num_weights(const orc_conv1d* c)
{
  return (long)c->K * c->in_ch * c->out_ch / c->groups + (c->has_bias ? c->out_ch : 0);
}

/* conv1d.cpp:11-56: for g, for i in out/g, for j in in/g, for k: W[k](g*opg+i, g*ipg+j); then bias. */
static const float* orc_conv1d_set_weights(orc_conv1d* c, const float* w)
{
  const int opg = c->out_ch / c->groups, ipg = c->in_ch / c->groups;
  memset(c->w, 0, sizeof(float) * (size_t)c->K * c->in_ch * c->out_ch);
  c->has_cached = 0;
  for (int g = 0; g < c->groups; g++)
    for (int i = 0; i < opg; i++)
      for (int j = 0; j < ipg; j++)
        for (int k = 0; k < c->K; k++)
          c->w[((size_t)k * c->out_ch + (g * opg + i)) * c->in_ch + (g * ipg + j)] = *(w++);
  for (int k = 0; k < c->K; k++)
    for (int o = 0; o < c->out_ch; o++)
      for (int i = 0; i < c->in_ch; i++)
        c->wt[((size_t)k * c->in_ch + i) * c->out_ch + o] = c->w[((size_t)k * c->out_ch + o) * c->in_ch + i];
  if (c->has_bias)
    for (int i = 0; i < c->out_ch; i++)
      c->bias[i] = *(w++);
  return w;
}

/* conv1d.cpp:128-149 */
static void orc_conv1d_set_max_buffer(orc_arena* ar, orc_conv1d* c, int max_buf)
{
  const long rf = c->K > 0 ? (long)(c->K - 1) * c->dilation : 0;
  orc_ring_reset(ar, &c->ring, c->in_ch, rf, max_buf);
  c->out = (float*)orc_alloc(ar, sizeof(float) * (size_t)c->out_ch * max_buf);
}

/* conv1d.cpp:163-183, 666-685, 768-775 */
static void orc_conv1d_process(orc_conv1d* c, const float* in, int in_stride, int num_frames)
{
  orc_ring_write(&c->ring, in, in_stride, num_frames);
  /* out = 0; for each tap k: out += W[k] * ring.Read(n, lookback_k) — conv1d.cpp:672-682 */
  for (int k = 0; k < c->K; k++)
  {
    const long lookback = (long)c->dilation * (c->K - 1 - k);
    const float* blk = orc_ring_read(&c->ring, lookback);
    const float* wk = c->wt + (size_t)k * c->out_ch * c->in_ch;
    if (k == 0)
    {
      /* 0 + (W[0] x) == W[0] x exactly, so the first tap can store instead of accumulate */
      orc_gemm(wk, blk, c->in_ch, c->in_ch, c->out_ch, num_frames, c->out, 0);
    }
    else
      orc_gemm(wk, blk, c->in_ch, c->in_ch, c->out_ch, num_frames, c->out, 1);
  }
  if (c->K == 0)
    memset(c->out, 0, sizeof(float) * (size_t)c->out_ch * num_frames);
  if (c->has_bias)
    for (int f = 0; f < num_frames; f++)
    {
      float* y = c->out + (size_t)f * c->out_ch;
      for (int o = 0; o < c->out_ch; o++)
        y[o] += c->bias[o];
    }
  c->ring.write_pos += num_frames; /* ring_buffer.cpp:59-62 Advance */
}

/* conv1d.cpp:157-161 + ring_buffer.cpp:64-69 */
static void orc_conv1d_cache_prewarm(orc_conv1d* c)
{
  memcpy(c->cached_col, c->ring.storage + (size_t)(c->ring.write_pos - 1) * c->in_ch, sizeof(float) * c->in_ch);
  c->has_cached = 1;
}

/* conv1d.cpp:151-155 + ring_buffer.cpp:71-76 */
static void orc_conv1d_prewarm_from_cache(orc_conv1d* c)
{
  for (long col = 0; col < c->ring.cols; col++)
    memcpy(c->ring.storage + (size_t)col * c->in_ch, c->cached_col, sizeof(float) * c->in_ch);
  c->ring.write_pos = c->ring.max_lookback;
}

/* ------------------------------------------------------------------------- */
/* FiLM — NAM/film.h:76-204                                                    */
/* ------------------------------------------------------------------------- */
typedef struct
{
  int active, do_shift, input_dim;
  orc_conv1x1 css; /* cond -> (shift?2:1)*input_dim, bias */
  float* out; /* [input_dim x max_buf] */
} orc_film;

static void orc_film_init(orc_arena* ar, orc_film* f, int active, int cond_dim, int input_dim, int shift, int groups)
{
  memset(f, 0, sizeof(*f));
  f->active = active;
  if (!active)
    return;
  f->do_shift = shift;
  f->input_dim = input_dim;
  orc_conv1x1_init(ar, &f->css, cond_dim, (shift ? 2 : 1) * input_dim, 1, groups);
}

/* film.h:76-190: out = in * scale (+ shift); scale = top rows, shift = bottom rows */
static void orc_film_process(orc_film* fl, const float* in, int in_stride, const float* cond, int cond_stride,
                             int num_frames)
{
  orc_conv1x1_process(&fl->css, cond, cond_stride, num_frames);
  const int D = fl->input_dim;
  const int ss_rows = fl->css.out_ch;
  for (int f = 0; f < num_frames; f++)
  {
    const float* x = in + (size_t)f * in_stride;
    const float* sc = fl->css.out + (size_t)f * ss_rows;
    float* y = fl->out + (size_t)f * D;
    if (fl->do_shift)
      for (int i = 0; i < D; i++)
        y[i] = x[i] * sc[i] + sc[D + i];
    else
      for (int i = 0; i < D; i++)
        y[i] = x[i] * sc[i];
  }
}

/* film.h:199-204: in-place variant (copy result back into `io`) */
static void orc_film_process_inplace(orc_film* fl, float* io, int io_stride, const float* cond, int cond_stride,
                                     int num_frames)
{
  orc_film_process(fl, io, io_stride, cond, cond_stride, num_frames);
  for (int f = 0; f < num_frames; f++)
    memcpy(io + (size_t)f * io_stride, fl->out + (size_t)f * fl->input_dim, sizeof(float) * fl->input_dim);
}

/* ------------------------------------------------------------------------- */
/* WaveNet Layer — NAM/wavenet/detail.h:44-155, NAM/wavenet/model.cpp:152-393  */
/* ------------------------------------------------------------------------- */
enum
{
  ORC_GATING_NONE = 0,
  ORC_GATING_GATED = 1,
  ORC_GATING_BLENDED = 2
};

enum
{
  FILM_CONV_PRE = 0,
  FILM_CONV_POST,
  FILM_MIXIN_PRE,
  FILM_MIXIN_POST,
  FILM_ACT_PRE,
  FILM_ACT_POST,
  FILM_LAYER1X1_POST,
  FILM_HEAD1X1_POST,
  FILM_COUNT
};

typedef struct
{
  int cond_size, channels, bottleneck, gating_mode;
  int has_layer1x1, has_head1x1;
  orc_conv1d conv;
  orc_conv1x1 mixin;
  orc_conv1x1 layer1x1;
  orc_conv1x1 head1x1;
  orc_act act, act2;
  orc_film film[FILM_COUNT];
  float* z; /* [zc x max_buf] */
  float* out_next; /* [channels x max_buf] */
  float* out_head; /* [head_out x max_buf] */
  int head_out_ch;
  int zc;
} orc_layer;

typedef struct
{
  int input_size, cond_size, head_size, head_kernel, head_dilation, head_bias;
  int channels, bottleneck, groups_input, groups_mixin;
  int layer1x1_active, layer1x1_groups, head1x1_active, head1x1_out, head1x1_groups;
  int film_cfg[FILM_COUNT][3]; /* active, shift, groups */
  int n_layers;
  orc_layer* layers;
  orc_conv1x1 rechannel;
  orc_conv1d head_rechannel;
  int head_output_size;
  float* layer_outputs; /* [channels x max_buf] */
  float* head_inputs; /* [head_output_size x max_buf] */
} orc_array;

#define ORC_MAX_ARRAYS 16
#define ORC_MAX_HEAD_CONVS 16

typedef struct orc_wavenet
{
  orc_arena cfg_arena; /* weights & structure */
  orc_arena buf_arena; /* buffers sized by max_buf (re-created on reset) */
  int in_channels, out_channels;
  int n_arrays;
  orc_array arrays[ORC_MAX_ARRAYS];
  float head_scale;
  struct orc_wavenet* condition_dsp; /* owned */
  /* post-stack head — model.cpp:21-103 */
  int with_head, head_n;
  orc_conv1d head_convs[ORC_MAX_HEAD_CONVS];
  orc_act head_act;
  float* scaled_head;
  int max_buf;
  int prewarm_samples;
  int finalized;
  float* cond_in; /* [in_channels x max_buf] */
  float* cond_out; /* [cond_dim x max_buf] */
  float* tmp_io; /* scratch for nested process */
} orc_wavenet;

static void orc_layer_build(orc_arena* ar, orc_layer* L, const orc_array* A, int kernel, int dilation,
                            const orc_act* act, int gating_mode, const orc_act* act2)
{
  /* detail.h:44-52 */
  memset(L, 0, sizeof(*L));
  L->cond_size = A->cond_size;
  L->channels = A->channels;
  L->bottleneck = A->bottleneck;
  L->gating_mode = gating_mode;
  L->zc = (gating_mode != ORC_GATING_NONE) ? 2 * A->bottleneck : A->bottleneck;
  orc_conv1d_init(ar, &L->conv, A->channels, L->zc, kernel, 1, dilation, A->groups_input);
  orc_conv1x1_init(ar, &L->mixin, A->cond_size, L->zc, 0, A->groups_mixin);
  L->act = *act;
  L->act2 = *act2;
  /* detail.h:54-85 */
  L->has_layer1x1 = A->layer1x1_active;
  if (L->has_layer1x1)
    orc_conv1x1_init(ar, &L->layer1x1, A->bottleneck, A->channels, 1, A->layer1x1_groups);
  L->has_head1x1 = A->head1x1_active;
  if (L->has_head1x1)
    orc_conv1x1_init(ar, &L->head1x1, A->bottleneck, A->head1x1_out, 1, A->head1x1_groups);
  L->head_out_ch = L->has_head1x1 ? A->head1x1_out : A->bottleneck;
  /* detail.h:103-154 FiLM dims */
  const int dims[FILM_COUNT] = {A->channels, L->zc, A->cond_size, L->zc, L->zc, A->bottleneck, A->channels,
                                A->head1x1_out};
  for (int i = 0; i < FILM_COUNT; i++)
  {
    int active = A->film_cfg[i][0];
    if (i == FILM_LAYER1X1_POST && !A->layer1x1_active)
      active = 0;
    if (i == FILM_HEAD1X1_POST && !A->head1x1_active)
      active = 0;
    orc_film_init(ar, &L->film[i], active, A->cond_size, dims[i], A->film_cfg[i][1], A->film_cfg[i][2]);
  }
}

/* model.cpp:152-181: conv, input_mixin, layer1x1?, head1x1?, then the 8 FiLMs in fixed order */
static const float* orc_layer_set_weights(orc_layer* L, const float* w)
{
  w = orc_conv1d_set_weights(&L->conv, w);
  w = orc_conv1x1_set_weights(&L->mixin, w);
  if (L->has_layer1x1)
    w = orc_conv1x1_set_weights(&L->layer1x1, w);
  if (L->has_head1x1)
    w = orc_conv1x1_set_weights(&L->head1x1, w);
  for (int i = 0; i < FILM_COUNT; i++)
    if (L->film[i].active)
      w = orc_conv1x1_set_weights(&L->film[i].css, w);
  return w;
}

static long orc_layer_num_weights(const orc_layer* L)
{
  long n = orc_conv1d_num_weights(&L->conv) + orc_conv1x1_num_weights(&L->mixin);
  if (L->has_layer1x1)
    n += orc_conv1x1_num_weights(&L->layer1x1);
  if (L->has_head1x1)
    n += orc_conv1x1_num_weights(&L->head1x1);
  for (int i = 0; i < FILM_COUNT; i++)
    if (L->film[i].active)
      n += orc_conv1x1_num_weights(&L->film[i].css);
  return n;
}

/* model.cpp:107-150 */
static void orc_layer_set_max_buffer(orc_arena* ar, orc_layer* L, int max_buf)
{
  orc_conv1d_set_max_buffer(ar, &L->conv, max_buf);
  orc_conv1x1_set_max_buffer(ar, &L->mixin, max_buf);
  L->z = (float*)orc_alloc(ar, sizeof(float) * (size_t)L->zc * max_buf);
  if (L->has_layer1x1)
    orc_conv1x1_set_max_buffer(ar, &L->layer1x1, max_buf);
  L->out_next = (float*)orc_alloc(ar, sizeof(float) * (size_t)L->channels * max_buf);
  L->out_head = (float*)orc_alloc(ar, sizeof(float) * (size_t)L->head_out_ch * max_buf);
  if (L->has_head1x1)
    orc_conv1x1_set_max_buffer(ar, &L->head1x1, max_buf);
  for (int i = 0; i < FILM_COUNT; i++)
    if (L->film[i].active)
    {
      orc_conv1x1_set_max_buffer(ar, &L->film[i].css, max_buf);
      L->film[i].out = (float*)orc_alloc(ar, sizeof(float) * (size_t)L->film[i].input_dim * max_buf);
    }
}

/* gating_activations.h:59-114 (GatingActivation::apply) */
static void orc_gating_apply(const orc_layer* L, float* z, int zc, int B, int num_frames)
{
  float a[256], g[256];
  for (int f = 0; f < num_frames; f++)
  {
    float* col = z + (size_t)f * zc;
    for (int c = 0; c < B; c++)
    {
      a[c] = col[c];
      g[c] = col[c + B];
    }
    orc_act_apply_column(&L->act, a, B);
    orc_act_apply_column(&L->act2, g, B);
    for (int c = 0; c < B; c++)
      col[c] = a[c] * g[c];
  }
}

/* gating_activations.h:165-228 (BlendingActivation::apply) */
static void orc_blending_apply(const orc_layer* L, float* z, int zc, int B, int num_frames)
{
  float pre[256], a[256], bl[256];
  for (int f = 0; f < num_frames; f++)
  {
    float* col = z + (size_t)f * zc;
    for (int c = 0; c < B; c++)
    {
      pre[c] = col[c];
      a[c] = col[c];
      bl[c] = col[c + B];
    }
    orc_act_apply_column(&L->act, a, B);
    orc_act_apply_column(&L->act2, bl, B);
    for (int c = 0; c < B; c++)
    {
      const float alpha = bl[c];
      col[c] = alpha * a[c] + (1.0f - alpha) * pre[c];
    }
  }
}

/* model.cpp:183-393 */
static void orc_layer_process(orc_layer* L, const float* input, const float* cond, int n)
{
  const int C = L->channels, B = L->bottleneck, zc = L->zc, cs = L->cond_size;
  /* Step 1: input convolution (+ pre/post FiLM) — model.cpp:189-203 */
  if (L->film[FILM_CONV_PRE].active)
  {
    orc_film_process(&L->film[FILM_CONV_PRE], input, C, cond, cs, n);
    orc_conv1d_process(&L->conv, L->film[FILM_CONV_PRE].out, C, n);
  }
  else
    orc_conv1d_process(&L->conv, input, C, n);
  if (L->film[FILM_CONV_POST].active)
    orc_film_process_inplace(&L->film[FILM_CONV_POST], L->conv.out, zc, cond, cs, n);
  /* input mixin — model.cpp:205-219 */
  if (L->film[FILM_MIXIN_PRE].active)
  {
    orc_film_process(&L->film[FILM_MIXIN_PRE], cond, cs, cond, cs, n);
    orc_conv1x1_process(&L->mixin, L->film[FILM_MIXIN_PRE].out, cs, n);
  }
  else
    orc_conv1x1_process(&L->mixin, cond, cs, n);
  if (L->film[FILM_MIXIN_POST].active)
    orc_film_process_inplace(&L->film[FILM_MIXIN_POST], L->mixin.out, zc, cond, cs, n);
  /* z = conv + mixin — model.cpp:220-221 */
  for (long i = 0; i < (long)zc * n; i++)
    L->z[i] = L->conv.out[i] + L->mixin.out[i];
  if (L->film[FILM_ACT_PRE].active)
    orc_film_process_inplace(&L->film[FILM_ACT_PRE], L->z, zc, cond, cs, n);

  /* Steps 2 & 3: activation + 1x1 — model.cpp:234-288 */
  if (L->gating_mode == ORC_GATING_NONE)
  {
    orc_act_apply_flat(&L->act, L->z, (long)zc * n);
    if (L->film[FILM_ACT_POST].active)
      orc_film_process_inplace(&L->film[FILM_ACT_POST], L->z, zc, cond, cs, n);
    if (L->has_layer1x1)
      orc_conv1x1_process(&L->layer1x1, L->z, zc, n);
  }
  else
  {
    if (L->gating_mode == ORC_GATING_GATED)
      orc_gating_apply(L, L->z, zc, B, n);
    else
      orc_blending_apply(L, L->z, zc, B, n);
    if (L->film[FILM_ACT_POST].active)
      orc_film_process_inplace(&L->film[FILM_ACT_POST], L->z, zc, cond, cs, n); /* top B rows, stride zc */
    if (L->has_layer1x1)
    {
      orc_conv1x1_process(&L->layer1x1, L->z, zc, n);
      /* quirk: layer1x1_post_film applied in the BLENDED branch only — model.cpp:282-286 */
      if (L->gating_mode == ORC_GATING_BLENDED && L->film[FILM_LAYER1X1_POST].active)
        orc_film_process_inplace(&L->film[FILM_LAYER1X1_POST], L->layer1x1.out, C, cond, cs, n);
    }
  }

  /* head output — model.cpp:290-352 */
  if (L->has_head1x1)
  {
    orc_conv1x1_process(&L->head1x1, L->z, zc, n);
    if (L->film[FILM_HEAD1X1_POST].active)
      orc_film_process_inplace(&L->film[FILM_HEAD1X1_POST], L->head1x1.out, L->head_out_ch, cond, cs, n);
    memcpy(L->out_head, L->head1x1.out, sizeof(float) * (size_t)L->head_out_ch * n);
  }
  else
  {
    for (int f = 0; f < n; f++)
      memcpy(L->out_head + (size_t)f * B, L->z + (size_t)f * zc, sizeof(float) * B);
  }

  /* residual — model.cpp:354-392 */
  if (L->has_layer1x1)
    for (long i = 0; i < (long)C * n; i++)
      L->out_next[i] = input[i] + L->layer1x1.out[i];
  else
    memcpy(L->out_next, input, sizeof(float) * (size_t)C * n);
}

/* ------------------------------------------------------------------------- */
/* LayerArray — model.cpp:397-569                                              */
/* ------------------------------------------------------------------------- */
static void orc_array_set_max_buffer(orc_arena* ar, orc_array* A, int max_buf)
{
  orc_conv1x1_set_max_buffer(ar, &A->rechannel, max_buf);
  orc_conv1d_set_max_buffer(ar, &A->head_rechannel, max_buf);
  for (int i = 0; i < A->n_layers; i++)
    orc_layer_set_max_buffer(ar, &A->layers[i], max_buf);
  A->layer_outputs = (float*)orc_alloc(ar, sizeof(float) * (size_t)A->channels * max_buf);
  A->head_inputs = (float*)orc_alloc(ar, sizeof(float) * (size_t)A->head_output_size * max_buf);
}

/* model.cpp:433-440 */
static long orc_array_receptive_field(const orc_array* A)
{
  long r = 0;
  for (int i = 0; i < A->n_layers; i++)
    r += (long)A->layers[i].conv.dilation * (A->layers[i].conv.K - 1);
  r += (long)A->head_rechannel.dilation * (A->head_rechannel.K - 1);
  return r;
}

/* model.cpp:463-549. head_in == NULL -> zero the accumulator (first array). */
static void orc_array_process(orc_array* A, const float* layer_inputs, const float* cond, const float* head_in, int n)
{
  const size_t hn = (size_t)A->head_output_size * n;
  if (head_in == NULL)
    memset(A->head_inputs, 0, sizeof(float) * hn);
  else
    memcpy(A->head_inputs, head_in, sizeof(float) * hn);
  orc_conv1x1_process(&A->rechannel, layer_inputs, A->input_size, n);
  const float* x = A->rechannel.out;
  for (int i = 0; i < A->n_layers; i++)
  {
    orc_layer_process(&A->layers[i], x, cond, n);
    for (size_t j = 0; j < hn; j++)
      A->head_inputs[j] += A->layers[i].out_head[j];
    x = A->layers[i].out_next;
  }
  memcpy(A->layer_outputs, x, sizeof(float) * (size_t)A->channels * n);
  orc_conv1d_process(&A->head_rechannel, A->head_inputs, A->head_output_size, n);
}

/* ------------------------------------------------------------------------- */
/* WaveNet — model.cpp:591-910                                                 */
/* ------------------------------------------------------------------------- */
ORC_API void* orc_wavenet_new(int in_channels, int with_head)
{
  orc_wavenet* wn = (orc_wavenet*)calloc(1, sizeof(orc_wavenet));
  wn->in_channels = in_channels;
  wn->with_head = with_head;
  return wn;
}

/* Array parameters as parsed from one entry of config["layers"] (model.cpp:932-1237). */
ORC_API int orc_wavenet_add_array(void* h, int input_size, int condition_size, int head_size, int head_kernel,
                                  int head_dilation, int head_bias, int channels, int bottleneck, int groups_input,
                                  int groups_mixin, int layer1x1_active, int layer1x1_groups, int head1x1_active,
                                  int head1x1_out, int head1x1_groups, const int* film_cfg /* [8][3] */, int n_layers)
{
  orc_wavenet* wn = (orc_wavenet*)h;
  if (wn->n_arrays >= ORC_MAX_ARRAYS)
    return -1;
  orc_array* A = &wn->arrays[wn->n_arrays];
  memset(A, 0, sizeof(*A));
  A->input_size = input_size;
  A->cond_size = condition_size;
  A->head_size = head_size;
  A->head_kernel = head_kernel;
  A->head_dilation = head_dilation;
  A->head_bias = head_bias;
  A->channels = channels;
  A->bottleneck = bottleneck;
  A->groups_input = groups_input;
  A->groups_mixin = groups_mixin;
  A->layer1x1_active = layer1x1_active;
  A->layer1x1_groups = layer1x1_groups;
  A->head1x1_active = head1x1_active;
  A->head1x1_out = head1x1_out;
  A->head1x1_groups = head1x1_groups;
  memcpy(A->film_cfg, film_cfg, sizeof(A->film_cfg));
  A->layers = (orc_layer*)orc_alloc(&wn->cfg_arena, sizeof(orc_layer) * (size_t)(n_layers > 0 ? n_layers : 1));
  A->n_layers = 0;
  /* model.cpp:397-401 */
  A->head_output_size = head1x1_active ? head1x1_out : bottleneck;
  orc_conv1x1_init(&wn->cfg_arena, &A->rechannel, input_size, channels, 0, 1);
  orc_conv1d_init(&wn->cfg_arena, &A->head_rechannel, A->head_output_size, head_size, head_kernel, head_bias ? 1 : 0,
                  head_dilation, 1);
  return wn->n_arrays++;
}

static void orc_act_from_cfg(orc_act* a, const float* cfg)
{
  /* cfg = [type, p0..p3, n_slopes, slopes...] */
  memset(a, 0, sizeof(*a));
  a->type = (int)cfg[0];
  for (int i = 0; i < 4; i++)
    a->p[i] = cfg[1 + i];
  a->n_slopes = (int)cfg[5];
  if (a->n_slopes > ORC_MAX_SLOPES)
    a->n_slopes = ORC_MAX_SLOPES;
  for (int i = 0; i < a->n_slopes; i++)
    a->slopes[i] = cfg[6 + i];
  if (a->type == ORC_ACT_PRELU && a->n_slopes == 0)
  {
    a->n_slopes = 1;
    a->slopes[0] = 0.01f;
  }
  if (a->type == ORC_ACT_LUT)
  {
    /* FastLUTActivation's constructor, activations.h:374-388: step = (max - min) / (size - 1); table[i] = f(min + i * step) */
    const int base = (int)a->p[2];
    a->lut_n = (int)a->p[3];
    a->lut_step = (a->p[1] - a->p[0]) / (float)(a->lut_n - 1);
    a->lut_inv_step = 1.0f / a->lut_step;
    a->lut = (float*)malloc(sizeof(float) * (size_t)a->lut_n);
    for (int i = 0; i < a->lut_n; i++)
    {
      const float x = a->p[0] + (float)i * a->lut_step;
      a->lut[i] = base == ORC_ACT_TANH ? tanhf(x) : base == ORC_ACT_SIGMOID ? orc_sigmoid(x) : x * orc_sigmoid(x);
    }
  }
}

ORC_API int orc_wavenet_add_layer(void* h, int array_idx, int kernel, int dilation, const float* act_cfg,
                                  int gating_mode, const float* act2_cfg)
{
  orc_wavenet* wn = (orc_wavenet*)h;
  orc_array* A = &wn->arrays[array_idx];
  orc_act a, a2;
  orc_act_from_cfg(&a, act_cfg);
  orc_act_from_cfg(&a2, act2_cfg);
  orc_layer_build(&wn->cfg_arena, &A->layers[A->n_layers], A, kernel, dilation, &a, gating_mode, &a2);
  return A->n_layers++;
}

/* Post-stack head: model.cpp:21-44. kernel_sizes[n]; channels; out_channels; activation. */
ORC_API int orc_wavenet_set_head(void* h, int in_ch, int channels, int out_channels, const int* kernel_sizes, int n,
                                 const float* act_cfg)
{
  orc_wavenet* wn = (orc_wavenet*)h;
  if (n > ORC_MAX_HEAD_CONVS)
    return -1;
  wn->head_n = n;
  orc_act_from_cfg(&wn->head_act, act_cfg);
  int cin = in_ch;
  for (int i = 0; i < n; i++)
  {
    const int cout = (i + 1 == n) ? out_channels : channels;
    orc_conv1d_init(&wn->cfg_arena, &wn->head_convs[i], cin, cout, kernel_sizes[i], 1, 1, 1);
    cin = cout;
  }
  return 0;
}

ORC_API void orc_wavenet_set_condition_dsp(void* h, void* cond)
{
  ((orc_wavenet*)h)->condition_dsp = (orc_wavenet*)cond;
}

ORC_API long orc_wavenet_expected_weights(void* h)
{
  orc_wavenet* wn = (orc_wavenet*)h;
  long n = 0;
  for (int a = 0; a < wn->n_arrays; a++)
  {
    orc_array* A = &wn->arrays[a];
    n += orc_conv1x1_num_weights(&A->rechannel);
    for (int i = 0; i < A->n_layers; i++)
      n += orc_layer_num_weights(&A->layers[i]);
    n += orc_conv1d_num_weights(&A->head_rechannel);
  }
  if (wn->with_head)
    for (int i = 0; i < wn->head_n; i++)
      n += orc_conv1d_num_weights(&wn->head_convs[i]);
  return n + 1; /* head_scale */
}

/* model.cpp:661-683 (+ :563-569). Returns 0 on exact consumption, -1 on mismatch. */
ORC_API int orc_wavenet_finalize(void* h, const float* weights, long n_weights)
{
  orc_wavenet* wn = (orc_wavenet*)h;
  if (orc_wavenet_expected_weights(h) != n_weights)
    return -1;
  const float* w = weights;
  for (int a = 0; a < wn->n_arrays; a++)
  {
    orc_array* A = &wn->arrays[a];
    w = orc_conv1x1_set_weights(&A->rechannel, w);
    for (int i = 0; i < A->n_layers; i++)
      w = orc_layer_set_weights(&A->layers[i], w);
    w = orc_conv1d_set_weights(&A->head_rechannel, w);
  }
  if (wn->with_head)
    for (int i = 0; i < wn->head_n; i++)
      w = orc_conv1d_set_weights(&wn->head_convs[i], w);
  wn->head_scale = *(w++); /* model.cpp:670: head_scale is the LAST weight, overriding the JSON field */
  if (w - weights != n_weights)
    return -1;
  /* out channels — model.cpp:578-586 */
  wn->out_channels = (wn->with_head && wn->head_n > 0) ? wn->head_convs[wn->head_n - 1].out_ch
                                                       : wn->arrays[wn->n_arrays - 1].head_size;
  /* prewarm samples — model.cpp:653-658 */
  wn->prewarm_samples = wn->condition_dsp ? wn->condition_dsp->prewarm_samples : 1;
  for (int a = 0; a < wn->n_arrays; a++)
    wn->prewarm_samples += (int)orc_array_receptive_field(&wn->arrays[a]);
  if (wn->with_head)
  {
    long rf = 1;
    for (int i = 0; i < wn->head_n; i++)
      rf += wn->head_convs[i].K - 1;
    wn->prewarm_samples += (int)(rf - 1);
  }
  wn->finalized = 1;
  return 0;
}

ORC_API int orc_wavenet_in_channels(void* h)
{
  return ((orc_wavenet*)h)->in_channels;
}
ORC_API int orc_wavenet_out_channels(void* h)
{
  return ((orc_wavenet*)h)->out_channels;
}
ORC_API int orc_wavenet_prewarm_samples(void* h)
{
  return ((orc_wavenet*)h)->prewarm_samples;
}
ORC_API float orc_wavenet_head_scale(void* h)
{
  return ((orc_wavenet*)h)->head_scale;
}

static int orc_wavenet_cond_dim(const orc_wavenet* wn)
{
  return wn->in_channels; /* model.h: _get_condition_dim() == NumInputChannels() */
}

/* model.cpp:685-728 — (re)allocates every buffer and zeroes every ring */
static void orc_wavenet_set_max_buffer(orc_wavenet* wn, int max_buf)
{
  orc_arena_free(&wn->buf_arena);
  wn->max_buf = max_buf;
  const int cd = orc_wavenet_cond_dim(wn);
  wn->cond_in = (float*)orc_alloc(&wn->buf_arena, sizeof(float) * (size_t)cd * max_buf);
  int cout = cd;
  if (wn->condition_dsp)
  {
    orc_wavenet_set_max_buffer(wn->condition_dsp, max_buf);
    cout = wn->condition_dsp->out_channels;
  }
  wn->cond_out = (float*)orc_alloc(&wn->buf_arena, sizeof(float) * (size_t)cout * max_buf);
  wn->tmp_io = (float*)orc_alloc(&wn->buf_arena, sizeof(float) * (size_t)(cd + cout + wn->out_channels) * max_buf);
  for (int a = 0; a < wn->n_arrays; a++)
    orc_array_set_max_buffer(&wn->buf_arena, &wn->arrays[a], max_buf);
  if (wn->with_head)
  {
    for (int i = 0; i < wn->head_n; i++)
      orc_conv1d_set_max_buffer(&wn->buf_arena, &wn->head_convs[i], max_buf);
    wn->scaled_head = (float*)orc_alloc(&wn->buf_arena, sizeof(float) * (size_t)wn->head_convs[0].in_ch * max_buf);
  }
}

/* `in`  : [in_channels][n]  planar (channel-major, like NAM_SAMPLE** input[ch][frame])
 * `out` : [out_channels][n] planar
 * model.cpp:822-910 */
ORC_API void orc_wavenet_process(void* h, const float* in, float* out, int n)
{
  orc_wavenet* wn = (orc_wavenet*)h;
  const int cd = orc_wavenet_cond_dim(wn);
  /* _set_condition_array — model.cpp:809-820 */
  for (int ch = 0; ch < cd; ch++)
    for (int j = 0; j < n; j++)
      wn->cond_in[(size_t)j * cd + ch] = in[(size_t)ch * n + j];
  /* _process_condition — model.cpp:777-807 */
  int cdim_out = cd;
  if (!wn->condition_dsp)
    memcpy(wn->cond_out, wn->cond_in, sizeof(float) * (size_t)cd * n);
  else
  {
    orc_wavenet* c = wn->condition_dsp;
    cdim_out = c->out_channels;
    float* tin = wn->tmp_io;
    float* tout = wn->tmp_io + (size_t)cd * n;
    /* float -> NAM_SAMPLE(double) -> float round trips are exact, so planar float copies suffice */
    memcpy(tin, in, sizeof(float) * (size_t)cd * n);
    orc_wavenet_process(c, tin, tout, n);
    for (int ch = 0; ch < cdim_out; ch++)
      for (int j = 0; j < n; j++)
        wn->cond_out[(size_t)j * cdim_out + ch] = tout[(size_t)ch * n + j];
  }
  /* layer arrays — model.cpp:832-850 */
  for (int a = 0; a < wn->n_arrays; a++)
  {
    if (a == 0)
      orc_array_process(&wn->arrays[a], wn->cond_in, wn->cond_out, NULL, n);
    else
      orc_array_process(&wn->arrays[a], wn->arrays[a - 1].layer_outputs, wn->cond_out,
                        wn->arrays[a - 1].head_rechannel.out, n);
  }
  const orc_array* last = &wn->arrays[wn->n_arrays - 1];
  const float* fh = last->head_rechannel.out;
  const int hs = last->head_size;
  if (wn->with_head)
  {
    /* model.cpp:854-883 + Head::process :86-103 */
    for (long i = 0; i < (long)hs * n; i++)
      wn->scaled_head[i] = wn->head_scale * fh[i];
    float* work = wn->scaled_head;
    for (int i = 0; i < wn->head_n; i++)
    {
      orc_conv1d* cv = &wn->head_convs[i];
      orc_act_apply_flat(&wn->head_act, work, (long)cv->in_ch * n);
      orc_conv1d_process(cv, work, cv->in_ch, n);
      work = cv->out;
    }
    const int oc = wn->out_channels;
    for (int ch = 0; ch < oc; ch++)
      for (int s = 0; s < n; s++)
        out[(size_t)ch * n + s] = work[(size_t)s * oc + ch];
    return;
  }
  /* model.cpp:887-909 */
  for (int ch = 0; ch < hs; ch++)
    for (int s = 0; s < n; s++)
      out[(size_t)ch * n + s] = wn->head_scale * fh[(size_t)s * hs + ch];
}

/* A long signal fed in `block`-frame process() calls, as tools/benchmodel.cpp:129-132 and
 * tools/render.cpp:146-197 do; the loop lives in C so timing it excludes any Python overhead. */
ORC_API void orc_wavenet_process_blocks(void* h, const float* in, float* out, long n_total, int block)
{
  orc_wavenet* wn = (orc_wavenet*)h;
  const int ic = wn->in_channels, oc = wn->out_channels;
  float* tin = (float*)malloc(sizeof(float) * (size_t)ic * block);
  float* tout = (float*)malloc(sizeof(float) * (size_t)oc * block);
  for (long s = 0; s < n_total; s += block)
  {
    const int n = (int)((n_total - s) < block ? (n_total - s) : block);
    for (int c = 0; c < ic; c++)
      memcpy(tin + (size_t)c * n, in + (size_t)c * n_total + s, sizeof(float) * n);
    orc_wavenet_process(wn, tin, tout, n);
    for (int c = 0; c < oc; c++)
      memcpy(out + (size_t)c * n_total + s, tout + (size_t)c * n, sizeof(float) * n);
  }
  free(tin);
  free(tout);
}

static int orc_wavenet_has_cache(const orc_wavenet* wn)
{
  /* model.cpp:749-757 */
  if (wn->condition_dsp)
    return 0;
  for (int a = 0; a < wn->n_arrays; a++)
  {
    const orc_array* A = &wn->arrays[a];
    if (!A->head_rechannel.has_cached)
      return 0;
    for (int i = 0; i < A->n_layers; i++)
      if (!A->layers[i].conv.has_cached)
        return 0;
  }
  for (int i = 0; wn->with_head && i < wn->head_n; i++)
    if (!wn->head_convs[i].has_cached)
      return 0;
  return 1;
}

/* DSP::prewarm dsp.cpp:67-101 + WaveNet::prewarm model.cpp:737-775 */
static void orc_wavenet_prewarm(orc_wavenet* wn)
{
  if (orc_wavenet_has_cache(wn))
  {
    for (int a = 0; a < wn->n_arrays; a++)
    {
      orc_array* A = &wn->arrays[a];
      for (int i = 0; i < A->n_layers; i++)
        orc_conv1d_prewarm_from_cache(&A->layers[i].conv);
      orc_conv1d_prewarm_from_cache(&A->head_rechannel);
    }
    for (int i = 0; wn->with_head && i < wn->head_n; i++)
      orc_conv1d_prewarm_from_cache(&wn->head_convs[i]);
    return;
  }
  const int bs = wn->max_buf > 0 ? wn->max_buf : 1;
  float* zin = (float*)calloc((size_t)wn->in_channels * bs, sizeof(float));
  float* zout = (float*)calloc((size_t)wn->out_channels * bs, sizeof(float));
  int done = 0;
  while (done < wn->prewarm_samples)
  {
    orc_wavenet_process(wn, zin, zout, bs);
    done += bs;
  }
  free(zin);
  free(zout);
  if (!wn->condition_dsp)
  {
    for (int a = 0; a < wn->n_arrays; a++)
    {
      orc_array* A = &wn->arrays[a];
      for (int i = 0; i < A->n_layers; i++)
        orc_conv1d_cache_prewarm(&A->layers[i].conv);
      orc_conv1d_cache_prewarm(&A->head_rechannel);
    }
    for (int i = 0; wn->with_head && i < wn->head_n; i++)
      orc_conv1d_cache_prewarm(&wn->head_convs[i]);
  }
}

/* DSP::Reset dsp.cpp:130-140 */
ORC_API void orc_wavenet_reset(void* h, int max_buf, int prewarm)
{
  orc_wavenet* wn = (orc_wavenet*)h;
  orc_wavenet_set_max_buffer(wn, max_buf);
  if (prewarm)
    orc_wavenet_prewarm(wn);
}

ORC_API void orc_wavenet_free(void* h)
{
  orc_wavenet* wn = (orc_wavenet*)h;
  if (!wn)
    return;
  if (wn->condition_dsp)
    orc_wavenet_free(wn->condition_dsp);
  orc_arena_free(&wn->buf_arena);
  orc_arena_free(&wn->cfg_arena);
  free(wn);
}

/* ------------------------------------------------------------------------- */
/* LSTM — NAM/lstm.cpp:9-168                                                   */
/* ------------------------------------------------------------------------- */
typedef struct
{
  int input_size, hidden;
  float* w; /* [4H][I+H] row-major */
  float* b; /* 4H */
  float* xh; /* I+H */
  float* ifgo; /* 4H */
  float* c; /* H */
} orc_lstm_cell;

typedef struct
{
  orc_arena arena;
  int in_ch, out_ch, n_layers, input_size, hidden;
  int fast; /* activations::Activation::using_fast_tanh */
  double sample_rate;
  orc_lstm_cell* cells;
  float* head_w; /* [out][H] */
  float* head_b;
  float* input;
  float* output;
  int max_buf;
} orc_lstm;

ORC_API long orc_lstm_expected_weights(int n_layers, int input_size, int hidden, int out_ch)
{
  long n = 0;
  for (int i = 0; i < n_layers; i++)
  {
    const int I = i == 0 ? input_size : hidden;
    n += 4L * hidden * (I + hidden) + 4L * hidden + 2L * hidden;
  }
  return n + (long)out_ch * hidden + out_ch;
}

/* lstm.cpp:9-29, 70-101 */
ORC_API void* orc_lstm_new(int in_ch, int out_ch, int n_layers, int input_size, int hidden, const float* weights,
                           long n_weights, double sample_rate, int fast_tanh)
{
  if (orc_lstm_expected_weights(n_layers, input_size, hidden, out_ch) != n_weights)
    return NULL;
  orc_lstm* m = (orc_lstm*)calloc(1, sizeof(orc_lstm));
  m->in_ch = in_ch;
  m->out_ch = out_ch;
  m->n_layers = n_layers;
  m->input_size = input_size;
  m->hidden = hidden;
  m->fast = fast_tanh;
  m->sample_rate = sample_rate;
  m->cells = (orc_lstm_cell*)orc_alloc(&m->arena, sizeof(orc_lstm_cell) * (size_t)(n_layers > 0 ? n_layers : 1));
  const float* w = weights;
  for (int l = 0; l < n_layers; l++)
  {
    orc_lstm_cell* c = &m->cells[l];
    const int I = l == 0 ? input_size : hidden, H = hidden;
    c->input_size = I;
    c->hidden = H;
    c->w = (float*)orc_alloc(&m->arena, sizeof(float) * (size_t)4 * H * (I + H));
    c->b = (float*)orc_alloc(&m->arena, sizeof(float) * 4 * H);
    c->xh = (float*)orc_alloc(&m->arena, sizeof(float) * (I + H));
    c->ifgo = (float*)orc_alloc(&m->arena, sizeof(float) * 4 * H);
    c->c = (float*)orc_alloc(&m->arena, sizeof(float) * H);
    for (int i = 0; i < 4 * H; i++)
      for (int j = 0; j < I + H; j++)
        c->w[(size_t)i * (I + H) + j] = *(w++);
    for (int i = 0; i < 4 * H; i++)
      c->b[i] = *(w++);
    for (int i = 0; i < H; i++)
      c->xh[I + i] = *(w++); /* initial hidden state */
    for (int i = 0; i < H; i++)
      c->c[i] = *(w++); /* initial cell state */
  }
  m->head_w = (float*)orc_alloc(&m->arena, sizeof(float) * (size_t)out_ch * hidden);
  m->head_b = (float*)orc_alloc(&m->arena, sizeof(float) * out_ch);
  for (int o = 0; o < out_ch; o++)
    for (int hh = 0; hh < hidden; hh++)
      m->head_w[(size_t)o * hidden + hh] = *(w++);
  for (int o = 0; o < out_ch; o++)
    m->head_b[o] = *(w++);
  m->input = (float*)orc_alloc(&m->arena, sizeof(float) * (input_size > in_ch ? input_size : in_ch));
  m->output = (float*)orc_alloc(&m->arena, sizeof(float) * out_ch);
  return m;
}

/* lstm.cpp:31-68 */
static void orc_lstm_cell_process(orc_lstm_cell* c, const float* x, int fast)
{
  const int H = c->hidden, I = c->input_size;
  for (int i = 0; i < I; i++)
    c->xh[i] = x[i];
  for (int r = 0; r < 4 * H; r++)
  {
    const float* wr = c->w + (size_t)r * (I + H);
    float sum = 0.0f;
    for (int j = 0; j < I + H; j++)
      sum += wr[j] * c->xh[j];
    c->ifgo[r] = sum + c->b[r];
  }
  const int io = 0, fo = H, go = 2 * H, oo = 3 * H;
  if (fast)
  {
    for (int i = 0; i < H; i++)
      c->c[i] = orc_fast_sigmoid(c->ifgo[i + fo]) * c->c[i]
                + orc_fast_sigmoid(c->ifgo[i + io]) * orc_fast_tanh(c->ifgo[i + go]);
    for (int i = 0; i < H; i++)
      c->xh[I + i] = orc_fast_sigmoid(c->ifgo[i + oo]) * orc_fast_tanh(c->c[i]);
  }
  else
  {
    for (int i = 0; i < H; i++)
      c->c[i] = orc_sigmoid(c->ifgo[i + fo]) * c->c[i] + orc_sigmoid(c->ifgo[i + io]) * tanhf(c->ifgo[i + go]);
    for (int i = 0; i < H; i++)
      c->xh[I + i] = orc_sigmoid(c->ifgo[i + oo]) * tanhf(c->c[i]);
  }
}

/* lstm.cpp:103-125, 136-168 */
ORC_API void orc_lstm_process(void* h, const float* in, float* out, int n)
{
  orc_lstm* m = (orc_lstm*)h;
  for (int f = 0; f < n; f++)
  {
    for (int ch = 0; ch < m->in_ch; ch++)
      m->input[ch] = in[(size_t)ch * n + f];
    if (m->n_layers == 0)
    {
      const int k = m->in_ch < m->out_ch ? m->in_ch : m->out_ch;
      for (int ch = 0; ch < k; ch++)
        m->output[ch] = m->input[ch];
      for (int ch = k; ch < m->out_ch; ch++)
        m->output[ch] = 0.0f;
    }
    else
    {
      orc_lstm_cell_process(&m->cells[0], m->input, m->fast);
      for (int l = 1; l < m->n_layers; l++)
        orc_lstm_cell_process(&m->cells[l], m->cells[l - 1].xh + m->cells[l - 1].input_size, m->fast);
      const orc_lstm_cell* last = &m->cells[m->n_layers - 1];
      const float* hs = last->xh + last->input_size;
      for (int o = 0; o < m->out_ch; o++)
      {
        float sum = 0.0f;
        for (int j = 0; j < m->hidden; j++)
          sum += m->head_w[(size_t)o * m->hidden + j] * hs[j];
        m->output[o] = sum + m->head_b[o];
      }
    }
    for (int ch = 0; ch < m->out_ch; ch++)
      out[(size_t)ch * n + f] = m->output[ch];
  }
}

ORC_API void orc_lstm_process_blocks(void* h, const float* in, float* out, long n_total, int block)
{
  orc_lstm* m = (orc_lstm*)h;
  const int ic = m->in_ch, oc = m->out_ch;
  float* tin = (float*)malloc(sizeof(float) * (size_t)ic * block);
  float* tout = (float*)malloc(sizeof(float) * (size_t)oc * block);
  for (long s = 0; s < n_total; s += block)
  {
    const int n = (int)((n_total - s) < block ? (n_total - s) : block);
    for (int c = 0; c < ic; c++)
      memcpy(tin + (size_t)c * n, in + (size_t)c * n_total + s, sizeof(float) * n);
    orc_lstm_process(m, tin, tout, n);
    for (int c = 0; c < oc; c++)
      memcpy(out + (size_t)c * n_total + s, tout + (size_t)c * n, sizeof(float) * n);
  }
  free(tin);
  free(tout);
}

/* lstm.cpp:127-134 */
ORC_API int orc_lstm_prewarm_samples(void* h)
{
  orc_lstm* m = (orc_lstm*)h;
  int r = (int)(0.5 * m->sample_rate);
  return r <= 0 ? 1 : r;
}

/* DSP::Reset + DSP::prewarm (dsp.cpp:67-101,130-140). NOTE: the LSTM state is NOT re-initialised by Reset. */
ORC_API void orc_lstm_reset(void* h, int max_buf, int prewarm)
{
  orc_lstm* m = (orc_lstm*)h;
  m->max_buf = max_buf;
  if (!prewarm)
    return;
  const int bs = max_buf > 0 ? max_buf : 1;
  float* zin = (float*)calloc((size_t)m->in_ch * bs, sizeof(float));
  float* zout = (float*)calloc((size_t)m->out_ch * bs, sizeof(float));
  const int target = orc_lstm_prewarm_samples(m);
  int done = 0;
  while (done < target)
  {
    orc_lstm_process(m, zin, zout, bs);
    done += bs;
  }
  free(zin);
  free(zout);
}

ORC_API void orc_lstm_free(void* h)
{
  orc_lstm* m = (orc_lstm*)h;
  if (!m)
    return;
  orc_arena_free(&m->arena);
  free(m);
}

/* ------------------------------------------------------------------------- */
/* Stand-alone primitive entry points (used by the KAT tests)                  */
/* ------------------------------------------------------------------------- */

/* Conv1D on a fresh (zero-history) instance, processing `n_calls` consecutive
 * blocks of `num_frames` frames each. in/out column-major [ch][total_frames]. */
ORC_API int orc_kat_conv1d(int in_ch, int out_ch, int K, int has_bias, int dilation, int groups, const float* weights,
                           long n_weights, const float* in, float* out, int num_frames, int n_calls, int max_buf)
{
  orc_arena ar = {0};
  orc_conv1d c;
  orc_conv1d_init(&ar, &c, in_ch, out_ch, K, has_bias, dilation, groups);
  if (orc_conv1d_num_weights(&c) != n_weights)
  {
    orc_arena_free(&ar);
    return -1;
  }
  orc_conv1d_set_weights(&c, weights);
  orc_conv1d_set_max_buffer(&ar, &c, max_buf);
  for (int i = 0; i < n_calls; i++)
  {
    orc_conv1d_process(&c, in + (size_t)i * num_frames * in_ch, in_ch, num_frames);
    memcpy(out + (size_t)i * num_frames * out_ch, c.out, sizeof(float) * (size_t)out_ch * num_frames);
  }
  orc_arena_free(&ar);
  return 0;
}

ORC_API int orc_kat_conv1x1(int in_ch, int out_ch, int has_bias, int groups, const float* weights, long n_weights,
                            const float* in, float* out, int num_frames)
{
  orc_arena ar = {0};
  orc_conv1x1 c;
  orc_conv1x1_init(&ar, &c, in_ch, out_ch, has_bias, groups);
  if (orc_conv1x1_num_weights(&c) != n_weights)
  {
    orc_arena_free(&ar);
    return -1;
  }
  orc_conv1x1_set_weights(&c, weights);
  orc_conv1x1_set_max_buffer(&ar, &c, num_frames);
  orc_conv1x1_process(&c, in, in_ch, num_frames);
  memcpy(out, c.out, sizeof(float) * (size_t)out_ch * num_frames);
  orc_arena_free(&ar);
  return 0;
}

ORC_API int orc_kat_film(int cond_dim, int input_dim, int shift, int groups, const float* weights, long n_weights,
                         const float* in, const float* cond, float* out, int num_frames)
{
  orc_arena ar = {0};
  orc_film f;
  orc_film_init(&ar, &f, 1, cond_dim, input_dim, shift, groups);
  if (orc_conv1x1_num_weights(&f.css) != n_weights)
  {
    orc_arena_free(&ar);
    return -1;
  }
  orc_conv1x1_set_weights(&f.css, weights);
  orc_conv1x1_set_max_buffer(&ar, &f.css, num_frames);
  f.out = (float*)orc_alloc(&ar, sizeof(float) * (size_t)input_dim * num_frames);
  orc_film_process(&f, in, input_dim, cond, cond_dim, num_frames);
  memcpy(out, f.out, sizeof(float) * (size_t)input_dim * num_frames);
  orc_arena_free(&ar);
  return 0;
}

ORC_API void orc_kat_activation(const float* act_cfg, float* data, long size)
{
  orc_act a;
  orc_act_from_cfg(&a, act_cfg);
  orc_act_apply_flat(&a, data, size);
}

/* mode: 1 = gated, 2 = blended. z is [2B x n] column-major; result in top B rows. */
ORC_API void orc_kat_gating(int mode, const float* act_cfg, const float* act2_cfg, int B, float* z, int num_frames)
{
  orc_layer L;
  memset(&L, 0, sizeof(L));
  orc_act_from_cfg(&L.act, act_cfg);
  orc_act_from_cfg(&L.act2, act2_cfg);
  if (mode == ORC_GATING_GATED)
    orc_gating_apply(&L, z, 2 * B, B, num_frames);
  else
    orc_blending_apply(&L, z, 2 * B, B, num_frames);
}

/* One WaveNet Layer in isolation (tools/test/test_wavenet/test_layer.cpp). Uses a
 * single-array single-layer wavenet handle built by the caller; exposes the
 * layer's next-layer output and head output for one Process call. */
ORC_API int orc_kat_layer(void* h, const float* weights, long n_weights, const float* input, const float* cond,
                          float* out_next, float* out_head, int num_frames)
{
  orc_wavenet* wn = (orc_wavenet*)h;
  orc_array* A = &wn->arrays[0];
  orc_layer* L = &A->layers[0];
  if (orc_layer_num_weights(L) != n_weights)
    return -1;
  orc_layer_set_weights(L, weights);
  orc_arena ar = {0};
  orc_layer_set_max_buffer(&ar, L, num_frames);
  orc_layer_process(L, input, cond, num_frames);
  memcpy(out_next, L->out_next, sizeof(float) * (size_t)L->channels * num_frames);
  memcpy(out_head, L->out_head, sizeof(float) * (size_t)L->head_out_ch * num_frames);
  orc_arena_free(&ar);
  return L->head_out_ch;
}
