// kernels.hip — hand-written HIP kernels for gfx950 (MI355X, CDNA4). No CUDA path, no shims.
//
// Mapping: the data-parallel axis is independent audio streams. For WaveNets one
// 64-lane wavefront owns one stream and walks its audio in blocks of 64 frames with LANE = FRAME,
// so every weight is wave-uniform (fetched by scalar loads, used as an SGPR operand of v_fma) and
// every history read is 64 consecutive floats. For LSTMs (a true recurrence) LANE = STREAM.
//
// Kernels:
//   nam_generic_kernel  interprets the op program of plan.h; covers every WaveNet feature the
//                       reference has (FiLM, gating/blending, grouped convs, head1x1, nested
//                       condition_dsp, post-stack head). Activations live in LDS rows.
//   nam_a1_kernel       register-resident specialisation for the plain A1 family
//                       (wavenet_a1_standard.nam): activations never leave VGPRs inside a layer array.
//   nam_lstm_kernel     LSTM, lanes = streams, h/c in LDS columns, I/O tiles transposed through LDS.
//
// Reference behaviour restated (file:line relative to the reference tree):
//   Layer::Process NAM/wavenet/model.cpp:183-393 · Conv1D::Process NAM/conv1d.cpp:163-183,666-685,768-775 ·
//   Conv1x1::process_ NAM/dsp.cpp:436-449,770-836 · activations NAM/activations.h:59-133 ·
//   gating NAM/gating_activations.h:59-228 · FiLM NAM/film.h:76-204 · LSTM NAM/lstm.cpp:31-168.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace namhip
{

// ------------------------------------------------------------------------------------------------
// Activations (NAM/activations.h:59-133)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float d_fast_tanh(const float x)
{
  const float ax = fabsf(x);
  const float x2 = x * x;
  return (x * (2.45550750702956f + 2.45550750702956f * ax + (0.893229853513558f + 0.821226666969744f * ax) * x2)
          / (2.44506634652299f + (2.44506634652299f + x2) * fabsf(x + 0.814642734961073f * x * ax)));
}
__device__ __forceinline__ float d_fast_sigmoid(const float x)
{
  return 0.5f * (d_fast_tanh(x * 0.5f) + 1.0f);
}
__device__ __forceinline__ float d_sigmoid(const float x)
{
  return 1.0f / (1.0f + expf(-x));
}

template <int TYPE>
__device__ __forceinline__ float d_act(float x, float p0, float p1, float p2, float p3, float slope)
{
  if constexpr (TYPE == ACT_TANH)
    return tanhf(x);
  else if constexpr (TYPE == ACT_HARDTANH)
  {
    const float t = x < -1.0f ? -1.0f : x;
    return t > 1.0f ? 1.0f : t;
  }
  else if constexpr (TYPE == ACT_FASTTANH)
    return d_fast_tanh(x);
  else if constexpr (TYPE == ACT_RELU)
    return x > 0.0f ? x : 0.0f;
  else if constexpr (TYPE == ACT_LEAKYRELU)
    return x > 0.0f ? x : p0 * x;
  else if constexpr (TYPE == ACT_PRELU)
    return x > 0.0f ? x : slope * x;
  else if constexpr (TYPE == ACT_SIGMOID)
    return d_sigmoid(x);
  else if constexpr (TYPE == ACT_SILU)
    return x * d_sigmoid(x);
  else if constexpr (TYPE == ACT_HARDSWISH)
  {
    const float t = x + 3.0f;
    const float c = t < 0.0f ? 0.0f : (t > 6.0f ? 6.0f : t);
    return x * c * (1.0f / 6.0f);
  }
  else if constexpr (TYPE == ACT_LEAKYHARDTANH)
  {
    if (x < p0)
      return (x - p0) * p2 + p0;
    else if (x > p1)
      return (x - p1) * p3 + p1;
    return x;
  }
  else if constexpr (TYPE == ACT_SOFTSIGN)
    return x / (1.0f + fabsf(x));
  else if constexpr (TYPE == ACT_FASTSIGMOID)
    return d_fast_sigmoid(x);
  else
    return x;
}

// run-time (wave-uniform) dispatch
__device__ __forceinline__ float d_act_rt(int type, float x, float p0, float p1, float p2, float p3, float slope)
{
  switch (type)
  {
    case ACT_TANH: return d_act<ACT_TANH>(x, p0, p1, p2, p3, slope);
    case ACT_HARDTANH: return d_act<ACT_HARDTANH>(x, p0, p1, p2, p3, slope);
    case ACT_FASTTANH: return d_act<ACT_FASTTANH>(x, p0, p1, p2, p3, slope);
    case ACT_RELU: return d_act<ACT_RELU>(x, p0, p1, p2, p3, slope);
    case ACT_LEAKYRELU: return d_act<ACT_LEAKYRELU>(x, p0, p1, p2, p3, slope);
    case ACT_PRELU: return d_act<ACT_PRELU>(x, p0, p1, p2, p3, slope);
    case ACT_SIGMOID: return d_act<ACT_SIGMOID>(x, p0, p1, p2, p3, slope);
    case ACT_SILU: return d_act<ACT_SILU>(x, p0, p1, p2, p3, slope);
    case ACT_HARDSWISH: return d_act<ACT_HARDSWISH>(x, p0, p1, p2, p3, slope);
    case ACT_LEAKYHARDTANH: return d_act<ACT_LEAKYHARDTANH>(x, p0, p1, p2, p3, slope);
    case ACT_SOFTSIGN: return d_act<ACT_SOFTSIGN>(x, p0, p1, p2, p3, slope);
    case ACT_FASTSIGMOID: return d_act<ACT_FASTSIGMOID>(x, p0, p1, p2, p3, slope);
    default: return x;
  }
}

__device__ __forceinline__ int uni(int v)
{
  return __builtin_amdgcn_readfirstlane(v);
}

// ------------------------------------------------------------------------------------------------
// Generic interpreter
// ------------------------------------------------------------------------------------------------
// One OP_CONV: dst[co][t] = (bias[co]) + sum_k sum_ci W[k][ci][co] * tap_k[ci][t]
// tap_k[ci][t] = src frame (t - L), L = (K-1-k)*dil: from the LDS block when t-L >= 0, else from the
// stream's history ring in HBM (frames of earlier blocks). Afterwards the block is appended to the ring.
template <int CB>
__device__ __forceinline__ void op_conv(const NamOp& op, float* lds, const float* __restrict__ blob, float* st,
                                        int* wpos_tbl, const int lane, const int nvalid)
{
  const float* src = lds + op.src;
  float* dst = lds + op.dst;
  const int cin = op.cin, cout = op.cout, cpad = op.cout_pad, K = op.k;
  const bool has_ring = op.state >= 0;
  const int R = op.ring;
  int wp = 0;
  float* ring = nullptr;
  if (has_ring)
  {
    wp = uni(wpos_tbl[op.ring_id]);
    ring = st + op.state;
  }
  for (int co0 = 0; co0 < cpad; co0 += CB)
  {
    float acc[CB];
#pragma unroll
    for (int j = 0; j < CB; j++)
      acc[j] = 0.0f;
    for (int k = 0; k < K; k++)
    {
      const int L = (K - 1 - k) * op.dil;
      const float* __restrict__ wk = blob + op.w + (size_t)k * cin * cpad + co0;
      if (L == 0)
      {
        for (int ci = 0; ci < cin; ci++)
        {
          const float x = src[ci * kBlock + lane];
#pragma unroll
          for (int j = 0; j < CB; j++)
            acc[j] = fmaf(wk[(size_t)ci * cpad + j], x, acc[j]);
        }
      }
      else if (L >= kBlock)
      {
        int idx = wp + lane - L;
        if (idx < 0)
          idx += R;
        for (int ci = 0; ci < cin; ci++)
        {
          const float x = ring[(size_t)ci * R + idx];
#pragma unroll
          for (int j = 0; j < CB; j++)
            acc[j] = fmaf(wk[(size_t)ci * cpad + j], x, acc[j]);
        }
      }
      else
      {
        const int tl = lane - L;
        const bool in_block = tl >= 0;
        int idx = wp + tl;
        if (idx < 0)
          idx += R;
        if (in_block)
          idx = 0; // keep the masked-off address in range
        const int lidx = in_block ? tl : 0;
        for (int ci = 0; ci < cin; ci++)
        {
          const float xl = src[ci * kBlock + lidx];
          const float xr = ring[(size_t)ci * R + idx];
          const float x = in_block ? xl : xr;
#pragma unroll
          for (int j = 0; j < CB; j++)
            acc[j] = fmaf(wk[(size_t)ci * cpad + j], x, acc[j]);
        }
      }
    }
    if (op.b >= 0)
    {
      const float* __restrict__ bias = blob + op.b + co0;
#pragma unroll
      for (int j = 0; j < CB; j++)
        acc[j] += bias[j];
    }
#pragma unroll
    for (int j = 0; j < CB; j++)
      if (co0 + j < cout)
        dst[(co0 + j) * kBlock + lane] = acc[j];
  }
  if (has_ring)
  {
    int widx = wp + lane;
    if (widx >= R)
      widx -= R;
    if (lane < nvalid)
      for (int ci = 0; ci < cin; ci++)
        ring[(size_t)ci * R + widx] = src[ci * kBlock + lane];
    int nwp = wp + nvalid;
    if (nwp >= R)
      nwp -= R;
    if (lane == 0)
      wpos_tbl[op.ring_id] = nwp;
  }
}

// `ops` and `blob` are separate `const __restrict__` kernel parameters (not struct members) so that
// the compiler can prove their loads are never clobbered by the state stores and lower the
// wave-uniform ones to scalar (s_load) instructions: weights then arrive as SGPR operands.
__global__ __launch_bounds__(64) void nam_generic_kernel(const NamOp* __restrict__ ops,
                                                         const float* __restrict__ blob, const GenericArgs a)
{
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  const int stream = a.stream_map ? a.stream_map[blockIdx.x] : (int)blockIdx.x;
  float* st = a.state + (size_t)stream * a.state_stride;
  int* wpos_tbl = reinterpret_cast<int*>(st);
  const float* in = a.in ? a.in + (size_t)stream * a.in_ch * a.io_stride : nullptr;
  float* out = a.out ? a.out + (size_t)stream * a.out_ch * a.io_stride : nullptr;

  for (int f0 = 0; f0 < a.n_frames; f0 += kBlock)
  {
    const int nvalid = min(kBlock, a.n_frames - f0);
    for (int pc = 0;; pc++)
    {
      const NamOp op = ops[pc];
      if (op.type == OP_END)
        break;
      switch (op.type)
      {
        case OP_LOAD_IN:
          for (int c = 0; c < op.cout; c++)
            lds[op.dst + c * kBlock + lane] = (in && lane < nvalid) ? in[(size_t)c * a.io_stride + f0 + lane] : 0.0f;
          break;
        case OP_STORE_OUT:
          if (out && lane < nvalid)
            for (int c = 0; c < op.cin; c++)
              out[(size_t)c * a.io_stride + f0 + lane] = lds[op.src + c * kBlock + lane];
          break;
        case OP_CONV:
          if (op.cb == 8)
            op_conv<8>(op, lds, blob, st, wpos_tbl, lane, nvalid);
          else
            op_conv<4>(op, lds, blob, st, wpos_tbl, lane, nvalid);
          break;
        case OP_FILM:
          for (int c = 0; c < op.cout; c++)
          {
            const float x = lds[op.src + c * kBlock + lane];
            const float sc = lds[op.aux + c * kBlock + lane];
            float y = x * sc;
            if (op.flag)
              y += lds[op.aux + (op.cout + c) * kBlock + lane];
            lds[op.dst + c * kBlock + lane] = y;
          }
          break;
        case OP_ACT:
        {
          const float p0 = blob[op.w], p1 = blob[op.w + 1], p2 = blob[op.w + 2], p3 = blob[op.w + 3];
          const int ns = op.ring;
          for (int c = 0; c < op.cout; c++)
          {
            float slope = 0.0f;
            if (op.k == ACT_PRELU)
            {
              // Activation::apply(float*, size) on column-major data: slopes[pos % n] (activations.h:283-297)
              const long pos = (long)(f0 + lane) * op.cout + c;
              slope = blob[op.w + 4 + (int)(pos % ns)];
            }
            const float x = lds[op.dst + c * kBlock + lane];
            lds[op.dst + c * kBlock + lane] = d_act_rt(op.k, x, p0, p1, p2, p3, slope);
          }
          break;
        }
        case OP_GATE:
        {
          // gating_activations.h:59-114 (gated) / :165-228 (blended); result in the top B rows
          const int B = op.cout;
          const float a0 = blob[op.w], a1 = blob[op.w + 1], a2 = blob[op.w + 2], a3 = blob[op.w + 3];
          const float g0 = blob[op.b], g1 = blob[op.b + 1], g2 = blob[op.b + 2], g3 = blob[op.b + 3];
          for (int c = 0; c < B; c++)
          {
            const float pre = lds[op.dst + c * kBlock + lane];
            const float gin = lds[op.dst + (c + B) * kBlock + lane];
            const float s1 = (op.k == ACT_PRELU) ? blob[op.w + 4 + c % op.ring] : 0.0f;
            const float s2 = (op.dil == ACT_PRELU) ? blob[op.b + 4 + c % op.ring_id] : 0.0f;
            const float av = d_act_rt(op.k, pre, a0, a1, a2, a3, s1);
            const float gv = d_act_rt(op.dil, gin, g0, g1, g2, g3, s2);
            lds[op.dst + c * kBlock + lane] = (op.flag == GATING_GATED) ? av * gv : gv * av + (1.0f - gv) * pre;
          }
          break;
        }
        case OP_ADD:
          for (int c = 0; c < op.cout; c++)
            lds[op.dst + c * kBlock + lane] = lds[op.src + c * kBlock + lane] + lds[op.aux + c * kBlock + lane];
          break;
        case OP_COPY:
          for (int c = 0; c < op.cout; c++)
            lds[op.dst + c * kBlock + lane] = lds[op.src + c * kBlock + lane];
          break;
        case OP_ZERO:
          for (int c = 0; c < op.cout; c++)
            lds[op.dst + c * kBlock + lane] = 0.0f;
          break;
        case OP_SCALE:
        {
          const float s = blob[op.w];
          for (int c = 0; c < op.cout; c++)
            lds[op.dst + c * kBlock + lane] = s * lds[op.src + c * kBlock + lane];
          break;
        }
        default: break;
      }
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------------
// A1-family register-resident kernel
// ------------------------------------------------------------------------------------------------
template <int C, int ACT>
__device__ __forceinline__ void a1_activate(float (&z)[C], float p0)
{
#pragma unroll
  for (int c = 0; c < C; c++)
    z[c] = d_act<ACT>(z[c], p0, 0.f, 0.f, 0.f, 0.f);
}

template <int C>
__device__ __forceinline__ void a1_activate_rt(float (&z)[C], int act, float p0)
{
  switch (act)
  {
    case ACT_TANH: a1_activate<C, ACT_TANH>(z, p0); break;
    case ACT_FASTTANH: a1_activate<C, ACT_FASTTANH>(z, p0); break;
    case ACT_HARDTANH: a1_activate<C, ACT_HARDTANH>(z, p0); break;
    case ACT_RELU: a1_activate<C, ACT_RELU>(z, p0); break;
    case ACT_LEAKYRELU: a1_activate<C, ACT_LEAKYRELU>(z, p0); break;
    case ACT_SIGMOID: a1_activate<C, ACT_SIGMOID>(z, p0); break;
    case ACT_SILU: a1_activate<C, ACT_SILU>(z, p0); break;
    case ACT_HARDSWISH: a1_activate<C, ACT_HARDSWISH>(z, p0); break;
    case ACT_SOFTSIGN: a1_activate<C, ACT_SOFTSIGN>(z, p0); break;
    default: break;
  }
}

// One layer array, lanes = frames. On entry:
//   win rows [0, in_size)  = layer_inputs (raw input for array 0, previous array's last-layer output otherwise)
//   hbuf rows [0, C)       = previous array's head output (ignored for the first array)
// On exit:
//   win rows [0, C)        = this array's last-layer output
//   hbuf rows [0, H)       = this array's head output (head rechannel applied)
template <int C>
__device__ __forceinline__ void a1_array(const A1Array* __restrict__ A, const float* __restrict__ blob, float* st,
                                         float* win, float* hbuf, const bool first, const float cond, const int wposv,
                                         const int lane, const int nvalid, const float act_p0)
{
  const int K = A->kernel, NL = A->n_layers, H = A->head_size, in_size = A->in_size, act = A->act;
  const float* __restrict__ w = blob + A->w_base;

  float x[C], head[C];
  // rechannel (Conv1x1, no bias) — model.cpp:492
#pragma unroll
  for (int co = 0; co < C; co++)
    x[co] = 0.0f;
  for (int ci = 0; ci < in_size; ci++)
  {
    const float v = win[ci * kBlock + lane];
#pragma unroll
    for (int co = 0; co < C; co++)
      x[co] = fmaf(w[ci * C + co], v, x[co]);
  }
  // head accumulator init — model.cpp:469 / :476-484
#pragma unroll
  for (int c = 0; c < C; c++)
    head[c] = first ? 0.0f : hbuf[c * kBlock + lane];
  w += in_size * C;

  for (int l = 0; l < NL; l++)
  {
    const int d = A->dil[l];
    const int R = A->ring_len[l];
    const int rid = A->ring_id[l];
    float* ring = st + A->ring_off[l];
    const float* __restrict__ cw = w;
    const float* __restrict__ cb = cw + K * C * C;
    const float* __restrict__ mx = cb + C;
    const float* __restrict__ w1 = mx + C;
    const float* __restrict__ b1 = w1 + C * C;
    w += A->layer_stride;

    const int wp = rid >= 0 ? __builtin_amdgcn_readlane(wposv, rid) : 0;

    // publish the layer input to the in-block window (LDS) and append it to the history ring (HBM)
    __syncthreads();
#pragma unroll
    for (int c = 0; c < C; c++)
      win[c * kBlock + lane] = x[c];
    if (rid >= 0)
    {
      int widx = wp + lane;
      if (widx >= R)
        widx -= R;
      if (lane < nvalid)
      {
#pragma unroll
        for (int c = 0; c < C; c++)
          ring[(size_t)c * R + widx] = x[c];
      }
    }
    __syncthreads();

    float acc[C];
#pragma unroll
    for (int c = 0; c < C; c++)
      acc[c] = 0.0f;

    // taps k = 0 .. K-2 look back L = (K-1-k)*d frames
    for (int k = 0; k < K - 1; k++)
    {
      const int L = (K - 1 - k) * d;
      const float* __restrict__ wk = cw + k * C * C;
      float xt[C];
      if (L >= kBlock)
      {
        int idx = wp + lane - L;
        if (idx < 0)
          idx += R;
#pragma unroll
        for (int c = 0; c < C; c++)
          xt[c] = ring[(size_t)c * R + idx];
      }
      else
      {
        const int tl = lane - L;
        const bool in_block = tl >= 0;
        int idx = wp + tl;
        if (idx < 0)
          idx += R;
        if (in_block)
          idx = 0;
        const int lidx = in_block ? tl : 0;
#pragma unroll
        for (int c = 0; c < C; c++)
        {
          const float xl = win[c * kBlock + lidx];
          const float xr = ring[(size_t)c * R + idx];
          xt[c] = in_block ? xl : xr;
        }
      }
#pragma unroll
      for (int ci = 0; ci < C; ci++)
#pragma unroll
        for (int co = 0; co < C; co++)
          acc[co] = fmaf(wk[ci * C + co], xt[ci], acc[co]);
    }
    // tap K-1: the current frame, straight from registers
    {
      const float* __restrict__ wk = cw + (K - 1) * C * C;
#pragma unroll
      for (int ci = 0; ci < C; ci++)
#pragma unroll
        for (int co = 0; co < C; co++)
          acc[co] = fmaf(wk[ci * C + co], x[ci], acc[co]);
    }
    // + bias, + input mixin (condition_size == 1), activation — model.cpp:220, :236
#pragma unroll
    for (int c = 0; c < C; c++)
      acc[c] = fmaf(mx[c], cond, acc[c] + cb[c]);
    a1_activate_rt<C>(acc, act, act_p0);
    // head accumulate — model.cpp:513-531
#pragma unroll
    for (int c = 0; c < C; c++)
      head[c] += acc[c];
    // layer1x1 + residual — model.cpp:241-244, :355-378
    float y[C];
#pragma unroll
    for (int c = 0; c < C; c++)
      y[c] = 0.0f;
#pragma unroll
    for (int ci = 0; ci < C; ci++)
#pragma unroll
      for (int co = 0; co < C; co++)
        y[co] = fmaf(w1[ci * C + co], acc[ci], y[co]);
#pragma unroll
    for (int c = 0; c < C; c++)
      x[c] = x[c] + (y[c] + b1[c]);
  }

  // last-layer output for the next array (model.cpp:536-545) and head rechannel (K = 1) (model.cpp:547-548)
  __syncthreads();
#pragma unroll
  for (int c = 0; c < C; c++)
    win[c * kBlock + lane] = x[c];
  const float* __restrict__ wh = w;
  const float* __restrict__ bh = wh + C * H;
  for (int h = 0; h < H; h++)
  {
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < C; c++)
      s = fmaf(wh[c * H + h], head[c], s);
    hbuf[h * kBlock + lane] = s + bh[h];
  }
  __syncthreads();
}

__global__ __launch_bounds__(64) void nam_a1_kernel(const A1Plan* __restrict__ P, const float* __restrict__ blob,
                                                    const A1Args a)
{
  __shared__ __attribute__((aligned(16))) float win[16 * kBlock];
  __shared__ __attribute__((aligned(16))) float hbuf[16 * kBlock];
  const int lane = threadIdx.x;
  const int stream = a.stream_map ? a.stream_map[blockIdx.x] : (int)blockIdx.x;
  float* st = a.state + (size_t)stream * a.state_stride;
  int* wpos_tbl = reinterpret_cast<int*>(st);
  const float* in = a.in ? a.in + (size_t)stream * a.io_stride : nullptr;
  float* out = a.out ? a.out + (size_t)stream * a.io_stride : nullptr;
  const int n_arrays = P->n_arrays;
  const int n_rings = P->n_rings;
  const float head_scale = blob[P->head_scale_off];

  for (int f0 = 0; f0 < a.n_frames; f0 += kBlock)
  {
    const int nvalid = min(kBlock, a.n_frames - f0);
    const float cond = (in && lane < nvalid) ? in[f0 + lane] : 0.0f;
    const int wposv = lane < n_rings ? wpos_tbl[lane] : 0;
    __syncthreads();
    win[lane] = cond;
    __syncthreads();
    for (int ai = 0; ai < n_arrays; ai++)
    {
      const A1Array* A = &P->arr[ai];
      const bool first = ai == 0;
      switch (A->channels)
      {
        case 16: a1_array<16>(A, blob, st, win, hbuf, first, cond, wposv, lane, nvalid, a.act_p0); break;
        case 12: a1_array<12>(A, blob, st, win, hbuf, first, cond, wposv, lane, nvalid, a.act_p0); break;
        case 8: a1_array<8>(A, blob, st, win, hbuf, first, cond, wposv, lane, nvalid, a.act_p0); break;
        case 6: a1_array<6>(A, blob, st, win, hbuf, first, cond, wposv, lane, nvalid, a.act_p0); break;
        case 4: a1_array<4>(A, blob, st, win, hbuf, first, cond, wposv, lane, nvalid, a.act_p0); break;
        case 3: a1_array<3>(A, blob, st, win, hbuf, first, cond, wposv, lane, nvalid, a.act_p0); break;
        case 2: a1_array<2>(A, blob, st, win, hbuf, first, cond, wposv, lane, nvalid, a.act_p0); break;
        case 1: a1_array<1>(A, blob, st, win, hbuf, first, cond, wposv, lane, nvalid, a.act_p0); break;
        default: break;
      }
    }
    if (out && lane < nvalid)
      out[f0 + lane] = head_scale * hbuf[lane];
    // advance every ring's write position by the frames consumed
    if (lane < n_rings)
    {
      const int R = P->ring_len_by_id[lane];
      int v = wposv + nvalid;
      if (v >= R)
        v -= R;
      wpos_tbl[lane] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// LSTM: lanes = streams (a true per-sample recurrence — NAM/lstm.cpp:103-168)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void nam_lstm_kernel(const float* __restrict__ blob, const LSTMArgs a)
{
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // LDS carve-up (floats): io tile [64 streams][65] | xh [(I+H) max][64] per layer | c [H][64] per layer | ifgo [4H][64]
  const int lane = threadIdx.x;
  const int s0 = blockIdx.x * kBlock;
  const int stream = s0 + lane;
  const bool live = stream < a.n_streams;
  const int H = a.hidden, NL = a.n_layers, I0 = a.input_size;
  const int in_ch = a.in_ch, out_ch = a.out_ch;
  float* tile_in = lds; // [in_ch][64][65]
  float* tile_out = tile_in + in_ch * kBlock * 65; // [out_ch][64][65]
  float* hs = tile_out + out_ch * kBlock * 65; // [NL][H][64]
  float* cs = hs + NL * H * kBlock; // [NL][H][64]
  float* ifgo = cs + NL * H * kBlock; // [4H][64]

  // load recurrent state
  float* st = a.state + (size_t)(live ? stream : 0) * a.state_stride;
  for (int l = 0; l < NL; l++)
    for (int i = 0; i < H; i++)
    {
      hs[(l * H + i) * kBlock + lane] = live ? st[(l * 2 + 0) * H + i] : 0.0f;
      cs[(l * H + i) * kBlock + lane] = live ? st[(l * 2 + 1) * H + i] : 0.0f;
    }

  for (int f0 = 0; f0 < a.n_frames; f0 += kBlock)
  {
    const int nvalid = min(kBlock, a.n_frames - f0);
    __syncthreads();
    // coalesced tile load: row r = stream s0+r, lane = frame
    for (int c = 0; c < in_ch; c++)
      for (int r = 0; r < kBlock; r++)
      {
        const int s = s0 + r;
        float v = 0.0f;
        if (a.in && s < a.n_streams && lane < nvalid)
          v = a.in[((size_t)s * in_ch + c) * a.io_stride + f0 + lane];
        tile_in[(c * kBlock + r) * 65 + lane] = v;
      }
    __syncthreads();
    for (int f = 0; f < nvalid; f++)
    {
      for (int l = 0; l < NL; l++)
      {
        const int I = l == 0 ? I0 : H;
        const float* __restrict__ W = blob + a.layer_w[l];
        const float* __restrict__ Bv = blob + a.layer_b[l];
        for (int r = 0; r < 4 * H; r++)
        {
          const float* __restrict__ wr = W + (size_t)r * (I + H);
          float sum = 0.0f;
          for (int j = 0; j < I; j++)
          {
            const float xv = (l == 0) ? tile_in[(j * kBlock + lane) * 65 + f] : hs[((l - 1) * H + j) * kBlock + lane];
            sum = fmaf(wr[j], xv, sum);
          }
          for (int j = 0; j < H; j++)
            sum = fmaf(wr[I + j], hs[(l * H + j) * kBlock + lane], sum);
          ifgo[r * kBlock + lane] = sum + Bv[r];
        }
        for (int i = 0; i < H; i++)
        {
          const float gi = ifgo[(i)*kBlock + lane], gf = ifgo[(i + H) * kBlock + lane];
          const float gg = ifgo[(i + 2 * H) * kBlock + lane], go = ifgo[(i + 3 * H) * kBlock + lane];
          const float cprev = cs[(l * H + i) * kBlock + lane];
          float cn, hn;
          if (a.fast)
          {
            cn = d_fast_sigmoid(gf) * cprev + d_fast_sigmoid(gi) * d_fast_tanh(gg);
            hn = d_fast_sigmoid(go) * d_fast_tanh(cn);
          }
          else
          {
            cn = d_sigmoid(gf) * cprev + d_sigmoid(gi) * tanhf(gg);
            hn = d_sigmoid(go) * tanhf(cn);
          }
          cs[(l * H + i) * kBlock + lane] = cn;
          hs[(l * H + i) * kBlock + lane] = hn;
        }
      }
      for (int o = 0; o < out_ch; o++)
      {
        const float* __restrict__ wr = blob + a.head_w + (size_t)o * H;
        float sum = 0.0f;
        for (int j = 0; j < H; j++)
          sum = fmaf(wr[j], hs[((NL - 1) * H + j) * kBlock + lane], sum);
        tile_out[(o * kBlock + lane) * 65 + f] = sum + blob[a.head_b + o];
      }
    }
    __syncthreads();
    if (a.out)
      for (int c = 0; c < out_ch; c++)
        for (int r = 0; r < kBlock; r++)
        {
          const int s = s0 + r;
          if (s < a.n_streams && lane < nvalid)
            a.out[((size_t)s * out_ch + c) * a.io_stride + f0 + lane] = tile_out[(c * kBlock + r) * 65 + lane];
        }
  }
  if (live)
    for (int l = 0; l < NL; l++)
      for (int i = 0; i < H; i++)
      {
        st[(l * 2 + 0) * H + i] = hs[(l * H + i) * kBlock + lane];
        st[(l * 2 + 1) * H + i] = cs[(l * H + i) * kBlock + lane];
      }
}

// ------------------------------------------------------------------------------------------------
// State initialisation
// ------------------------------------------------------------------------------------------------
__global__ void nam_fill_state_kernel(float* state, long state_stride, const int* stream_map, int n_streams,
                                      const float* init, int n_init, int state_floats)
{
  // one block per stream; copies `init` (n_init floats) then zero-fills the rest
  const int stream = stream_map ? stream_map[blockIdx.x] : (int)blockIdx.x;
  float* st = state + (size_t)stream * state_stride;
  for (int i = threadIdx.x; i < state_floats; i += blockDim.x)
    st[i] = (init && i < n_init) ? init[i] : 0.0f;
}

// ------------------------------------------------------------------------------------------------
// Launch wrappers (host)
// ------------------------------------------------------------------------------------------------
hipError_t launch_generic(const GenericArgs& a, int n_blocks, int lds_bytes, hipStream_t stream)
{
  if (lds_bytes > 64 * 1024)
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(nam_generic_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess)
      return e;
  }
  hipLaunchKernelGGL(nam_generic_kernel, dim3(n_blocks), dim3(64), lds_bytes, stream, a.ops, a.blob, a);
  return hipGetLastError();
}

hipError_t launch_a1(const A1Args& a, int n_blocks, hipStream_t stream)
{
  hipLaunchKernelGGL(nam_a1_kernel, dim3(n_blocks), dim3(64), 0, stream, a.plan, a.blob, a);
  return hipGetLastError();
}

int lstm_lds_bytes(const LSTMArgs& a)
{
  const int floats = (a.in_ch + a.out_ch) * kBlock * 65 + 2 * a.n_layers * a.hidden * kBlock + 4 * a.hidden * kBlock;
  return floats * (int)sizeof(float);
}

hipError_t launch_lstm(const LSTMArgs& a, hipStream_t stream)
{
  const int lds_bytes = lstm_lds_bytes(a);
  if (lds_bytes > 64 * 1024)
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(nam_lstm_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess)
      return e;
  }
  const int n_blocks = (a.n_streams + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(nam_lstm_kernel, dim3(n_blocks), dim3(64), lds_bytes, stream, a.blob, a);
  return hipGetLastError();
}

hipError_t launch_fill_state(float* state, long state_stride, const int* stream_map, int n_streams, const float* init,
                             int n_init, int state_floats, hipStream_t stream)
{
  hipLaunchKernelGGL(nam_fill_state_kernel, dim3(n_streams), dim3(256), 0, stream, state, state_stride, stream_map,
                     n_streams, init, n_init, state_floats);
  return hipGetLastError();
}

} // namespace namhip
