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
//   nam_a1_mfma_kernel  the headline kernel: fp32 MFMA, 4 compute + 4 mover wavefronts per stream (kernel size 3).
//   nam_kt_mfma_kernel  fp32 MFMA for single-array models with any per-layer kernel size (A2), 4 wavefronts per stream.
//   nam_lstm_kernel     LSTM, lanes = streams, h/c in LDS columns, I/O tiles transposed through LDS.
//   nam_lstm_mfma_kernel / nam_lstm_mfma_reg_kernel   LSTM on MFMA, 16 streams per wavefront.
//
// Reference behaviour restated (file:line relative to the reference tree):
//   Layer::Process NAM/wavenet/model.cpp:183-393 · Conv1D::Process NAM/conv1d.cpp:163-183,666-685,768-775 ·
//   Conv1x1::process_ NAM/dsp.cpp:436-449,770-836 · activations NAM/activations.h:59-133 ·
//   gating NAM/gating_activations.h:59-228 · FiLM NAM/film.h:76-204 · LSTM NAM/lstm.cpp:31-168.
#include <hip/hip_runtime.h>

#include <type_traits>

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
using mf_f4 = __attribute__((ext_vector_type(4))) float;

// One OP_CONV: dst[co][t] = (bias[co]) + sum_k sum_ci W[k][ci][co] * tap_k[ci][t]
// tap_k[ci][t] = src frame (t - L), L = (K-1-k)*dil: from the LDS block when t-L >= 0, else from the
// stream's history ring in HBM (frames of earlier blocks). Afterwards the block is appended to the ring.
template <int CB, bool WLDS>
__device__ __forceinline__ void op_conv(const NamOp& op, float* lds, const float* __restrict__ blob, const float* wlds,
                                        float* st, int* wpos_tbl, const int lane, const int nvalid)
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
      const float* wkl = wlds + op.w + (size_t)k * cin * cpad + co0; // the same weights in LDS (WLDS)
      // acc[j] += W[k][ci][co0 + j] * x: weights as SGPR operands (scalar loads) or, with WLDS, as broadcast
      // 16-byte LDS reads
      auto fma_row = [&](int ci, float x) {
        if constexpr (WLDS)
        {
          float wv[CB];
#pragma unroll
          for (int j = 0; j < CB; j += 4)
          {
            const mf_f4 q = *reinterpret_cast<const mf_f4*>(wkl + (size_t)ci * cpad + j);
            wv[j] = q[0], wv[j + 1] = q[1], wv[j + 2] = q[2], wv[j + 3] = q[3];
          }
#pragma unroll
          for (int j = 0; j < CB; j++)
            acc[j] = fmaf(wv[j], x, acc[j]);
        }
        else
        {
#pragma unroll
          for (int j = 0; j < CB; j++)
            acc[j] = fmaf(wk[(size_t)ci * cpad + j], x, acc[j]);
        }
      };
      if (L == 0)
      {
        for (int ci = 0; ci < cin; ci++)
          fma_row(ci, src[ci * kBlock + lane]);
      }
      else if (L >= kBlock)
      {
        int idx = wp + lane - L;
        if (idx < 0)
          idx += R;
        for (int ci = 0; ci < cin; ci++)
          fma_row(ci, ring[(size_t)idx * cin + ci]);
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
          const float xr = ring[(size_t)idx * cin + ci];
          fma_row(ci, in_block ? xl : xr);
        }
      }
    }
    if (op.b >= 0)
    {
      const float* __restrict__ bias = blob + op.b + co0;
#pragma unroll
      for (int j = 0; j < CB; j++)
        acc[j] += WLDS ? wlds[op.b + co0 + j] : bias[j];
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
        ring[(size_t)widx * cin + ci] = src[ci * kBlock + lane];
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
template <bool WLDS>
__global__ __launch_bounds__(64) void nam_generic_kernel(const NamOp* __restrict__ ops,
                                                         const float* __restrict__ blob, const GenericArgs a)
{
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  const float* wlds = lds + a.w_lds_off;
  if constexpr (WLDS)
  {
    float* wdst = lds + a.w_lds_off;
    for (int i = lane * 4; i < a.blob_floats; i += 64 * 4)
      *reinterpret_cast<mf_f4*>(wdst + i) = *reinterpret_cast<const mf_f4*>(blob + i);
    __syncthreads();
  }
  const int stream = a.stream_map ? a.stream_map[blockIdx.x] : (int)blockIdx.x;
  float* st = a.state + (size_t)stream * a.state_stride;
  int* wpos_tbl = reinterpret_cast<int*>(st);
  const float* in = a.in ? a.in + (size_t)stream * a.in_ch * a.io_stride : nullptr;
  float* out = a.out ? a.out + (size_t)stream * a.out_ch * a.io_stride : nullptr;

  for (int f0 = 0; f0 < a.n_frames; f0 += kBlock)
  {
    const int nvalid = min(kBlock, a.n_frames - f0);
    // the descriptor of the op after this one is requested before this op runs (the program ends with OP_END and
    // plan.cpp pads it with one more so that pc + 1 is always readable)
    NamOp next_op = ops[0];
    for (int pc = 0;; pc++)
    {
      const NamOp op = next_op;
      if (op.type == OP_END)
        break;
      next_op = ops[pc + 1];
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
            op_conv<8, WLDS>(op, lds, blob, wlds, st, wpos_tbl, lane, nvalid);
          else
            op_conv<4, WLDS>(op, lds, blob, wlds, st, wpos_tbl, lane, nvalid);
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
  const int NL = A->n_layers, H = A->head_size, in_size = A->in_size, act = A->act;
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

  // taps k = 0 .. K-2 of a dilated causal conv over `cur` (this lane's frame, in registers; already published to
  // LDS rows `lw` and appended to `ring`): acc[co] += sum_k sum_ci wk[k][ci][co] * in[ci][t - (K-1-k) d]
  auto shifted_taps = [&](float (&acc)[C], const float* __restrict__ cw, int K, int d, const float* lw, const float* ring,
                          int R, int wp) {
    // kTapChunk taps at a time: every load of the chunk (ring rows in HBM / L2, window rows in LDS) is issued
    // before the first FMA needs one, so a chunk costs one memory round trip instead of one per tap
    constexpr int kTapChunk = 4;
    for (int k0 = 0; k0 < K - 1; k0 += kTapChunk)
    {
      float xt[kTapChunk][C];
#pragma unroll
      for (int u = 0; u < kTapChunk; u++)
      {
        const int k = min(k0 + u, K - 2);
        const int L = (K - 1 - k) * d;
        const int tl = lane - L;
        const bool in_block = tl >= 0;
        int idx = wp + tl;
        if (idx < 0)
          idx += R;
        if (in_block)
          idx = 0; // keep the masked-off address in range
        const int lidx = in_block ? tl : 0;
#pragma unroll
        for (int c = 0; c < C; c++)
        {
          const float xl = lw[c * kBlock + lidx];
          const float xr = ring[(size_t)idx * C + c];
          xt[u][c] = in_block ? xl : xr;
        }
      }
#pragma unroll
      for (int u = 0; u < kTapChunk; u++)
        if (k0 + u < K - 1)
        {
          const float* __restrict__ wk = cw + (k0 + u) * C * C;
#pragma unroll
          for (int ci = 0; ci < C; ci++)
#pragma unroll
            for (int co = 0; co < C; co++)
              acc[co] = fmaf(wk[ci * C + co], xt[u][ci], acc[co]);
        }
    }
  };
  // publish this lane's frame of `v` to LDS rows `lw` and append it to the history ring
  auto publish = [&](const float (&v)[C], float* lw, float* ring, int R, int wp, bool has_ring) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < C; c++)
      lw[c * kBlock + lane] = v[c];
    if (has_ring)
    {
      int widx = wp + lane;
      if (widx >= R)
        widx -= R;
      if (lane < nvalid)
      {
#pragma unroll
        for (int c = 0; c < C; c++)
          ring[(size_t)widx * C + c] = v[c];
      }
    }
    __syncthreads();
  };

  for (int l = 0; l < NL; l++)
  {
    const int K = A->ksize[l];
    const int d = A->dil[l];
    const int R = A->ring_len[l];
    const int rid = A->ring_id[l];
    float* ring = st + A->ring_off[l];
    const float* __restrict__ cw = w + A->layer_off[l];
    const float* __restrict__ cb = cw + K * C * C;
    const float* __restrict__ mx = cb + C;
    const float* __restrict__ w1 = mx + C;
    const float* __restrict__ b1 = w1 + C * C;

    const int wp = rid >= 0 ? __builtin_amdgcn_readlane(wposv, rid) : 0;

    // publish the layer input to the in-block window (LDS) and append it to the history ring (HBM)
    publish(x, win, ring, R, wp, rid >= 0);

    float acc[C];
#pragma unroll
    for (int c = 0; c < C; c++)
      acc[c] = 0.0f;
    shifted_taps(acc, cw, K, d, win, ring, R, wp);
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
    a1_activate_rt<C>(acc, act, A->act_p0);
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

  // head rechannel: a Conv1D over the head accumulator (model.cpp:399-400, 547-548). K = 1: a plain 1x1 into any
  // number of output channels. K > 1 (A2: 16 taps): a single output channel (plan.cpp); the accumulator goes
  // through hbuf / its own ring exactly like a layer input.
  const int KH = A->head_k;
  const float* __restrict__ wh = w + A->head_off;
  const float* __restrict__ bh = wh + KH * C * H;
  float hout0 = 0.0f;
  if (KH > 1)
  {
    const int hrid = A->head_ring_id;
    const int hwp = hrid >= 0 ? __builtin_amdgcn_readlane(wposv, hrid) : 0;
    float* hring = st + A->head_ring_off;
    publish(head, hbuf, hring, A->head_ring_len, hwp, hrid >= 0);
    // with H == 1 the packed taps [k][c][1] are [k][c]: reuse the C-wide tap routine on a C x C view whose column 0
    // is the real one would waste C x the FMAs; do the single output directly
    float s = 0.0f;
    for (int k = 0; k < KH; k++)
    {
      const int L = (KH - 1 - k) * A->head_dil;
      const int tl = lane - L;
      const bool in_block = tl >= 0;
      int idx = hwp + tl;
      if (idx < 0)
        idx += A->head_ring_len;
      if (in_block)
        idx = 0;
      const int lidx = in_block ? tl : 0;
#pragma unroll
      for (int c = 0; c < C; c++)
      {
        const float xl = hbuf[c * kBlock + lidx];
        const float xr = (L > 0) ? hring[(size_t)idx * C + c] : 0.0f;
        s = fmaf(wh[k * C + c], (in_block || L == 0) ? xl : xr, s);
      }
    }
    hout0 = s + bh[0];
  }
  // last-layer output for the next array (model.cpp:536-545)
  __syncthreads();
#pragma unroll
  for (int c = 0; c < C; c++)
    win[c * kBlock + lane] = x[c];
  if (KH > 1)
    hbuf[lane] = hout0;
  else
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
// A1-family MFMA kernel — shared helpers (the kernel itself, nam_a1_mfma_kernel, is further down)
// ------------------------------------------------------------------------------------------------
// Every matrix product of the model is (C x Kdim) * (Kdim x 64 frames); compute wave w owns frames
// [16w, 16w+16) and issues v_mfma_f32_16x16x4_f32 (exact fp32: bitwise an ordered fmaf chain).
// Lane l = (g = l >> 4, j = l & 15) of wave w, FULL layout (plan.cpp describes the HALF layout of
// 8-channel arrays):
//   D (4 VGPR)  out channels 4g + r, r = 0..3, of frame 16w + j          (residual x, head, z live here)
//   B operand   k-step s feeds row k = g with channel 4g + s of frame 16w + j — THE LANE'S OWN D VALUES,
//               so the current tap, the 1x1, the rechannel and the head need no data movement at all
//   A operand   tile value W[out = j][in = 4g + s] (plan.cpp packs the tiles for exactly this mapping)
// Only the time-shifted taps leave the registers: each lane fetches its channels of frame
// (16w + j - L) with ONE LDS read from a frame-major window (lookback L <= 64: [previous 64 |
// current 64] frames) or tap buffer (L > 64), both filled from the stream's frame-major history ring
// in HBM with 16-byte accesses.
namespace mf
{
using f4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ float rcp(float x)
{
  return __builtin_amdgcn_rcpf(x);
}
// tanh(x) = 1 - 2 / (exp(2x) + 1) on the hardware exp2 / rcp units (abs error ~1e-7)
__device__ __forceinline__ float tanh_hw(float x)
{
  const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f); // exp(2x) = 2^(2x*log2(e))
  return 1.0f - 2.0f * rcp(e + 1.0f);
}
__device__ __forceinline__ float fast_tanh_hw(const float x)
{
  const float ax = fabsf(x);
  const float x2 = x * x;
  const float num = x * (2.45550750702956f + 2.45550750702956f * ax + (0.893229853513558f + 0.821226666969744f * ax) * x2);
  const float den = 2.44506634652299f + (2.44506634652299f + x2) * fabsf(x + 0.814642734961073f * x * ax);
  return num * rcp(den);
}
__device__ __forceinline__ float sigmoid_hw(float x)
{
  return rcp(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
// fast_sigmoid of the reference's LSTM (lstm.cpp:48-58: 0.5 (fast_tanh(x / 2) + 1)) on the hardware rcp
__device__ __forceinline__ float fast_sigmoid_hw(const float x)
{
  return 0.5f * (fast_tanh_hw(x * 0.5f) + 1.0f);
}
__device__ __forceinline__ float act_hw(int type, float x, float p0)
{
  switch (type)
  {
    case ACT_TANH: return tanh_hw(x);
    case ACT_FASTTANH: return fast_tanh_hw(x);
    case ACT_HARDTANH: return fminf(fmaxf(x, -1.0f), 1.0f);
    case ACT_RELU: return x > 0.0f ? x : 0.0f;
    case ACT_LEAKYRELU: return x > 0.0f ? x : p0 * x;
    case ACT_SIGMOID: return sigmoid_hw(x);
    case ACT_SILU: return x * sigmoid_hw(x);
    case ACT_HARDSWISH:
    {
      const float t = fminf(fmaxf(x + 3.0f, 0.0f), 6.0f);
      return x * t * (1.0f / 6.0f);
    }
    case ACT_SOFTSIGN: return x * rcp(1.0f + fabsf(x));
    default: return x;
  }
}
// whole-vector activation; ACT_T >= 0 resolves the type at compile time (the two kernels that matter:
// Fasttanh = benchmodel default, Tanh), ACT_T < 0 dispatches once per job on the run-time type
template <int ACT_T>
__device__ __forceinline__ f4 act4(int type, const f4& v, float p0)
{
  f4 r;
  if constexpr (ACT_T == ACT_FASTTANH)
  {
#pragma unroll
    for (int i = 0; i < 4; i++)
      r[i] = fast_tanh_hw(v[i]);
  }
  else if constexpr (ACT_T == ACT_TANH)
  {
#pragma unroll
    for (int i = 0; i < 4; i++)
      r[i] = tanh_hw(v[i]);
  }
  else if constexpr (ACT_T == ACT_LEAKYRELU)
  {
#pragma unroll
    for (int i = 0; i < 4; i++)
      r[i] = v[i] > 0.0f ? v[i] : p0 * v[i];
  }
  else
  {
    // one dispatch per vector, not per element (a 10-way compare cascade per element costs hundreds of cycles on a
    // lone wavefront)
#define NAM_ACT4_CASE(T) \
  case T: \
    _Pragma("unroll") for (int i = 0; i < 4; i++) r[i] = act_hw(T, v[i], p0); \
    break;
    switch (type)
    {
      NAM_ACT4_CASE(ACT_TANH)
      NAM_ACT4_CASE(ACT_FASTTANH)
      NAM_ACT4_CASE(ACT_HARDTANH)
      NAM_ACT4_CASE(ACT_RELU)
      NAM_ACT4_CASE(ACT_LEAKYRELU)
      NAM_ACT4_CASE(ACT_SIGMOID)
      NAM_ACT4_CASE(ACT_SILU)
      NAM_ACT4_CASE(ACT_HARDSWISH)
      NAM_ACT4_CASE(ACT_SOFTSIGN)
      default: r = v; break;
    }
#undef NAM_ACT4_CASE
  }
  return r;
}
// workgroup barrier that orders LDS traffic only: outstanding global loads stay in flight
__device__ __forceinline__ void lds_barrier()
{
  // the builtin, not inline asm: hipcc's waitcnt pass then knows nothing is outstanding on lgkmcnt after the
  // barrier; with an asm wait it re-waits (lgkmcnt(0)) before the first use of any register loaded in the previous job,
  // i.e. in front of the own-tap MFMA chain that is there to cover the latency of the shifted-tap reads
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0); vmcnt / expcnt untouched
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
template <int NK>
__device__ __forceinline__ f4 mfma_n(const f4& a, const f4& b, f4 acc)
{
#pragma unroll
  for (int s = 0; s < NK; s++)
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc, 0, 0, 0);
  return acc;
}
using f2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ f4 lds_ld4(const char* lds, unsigned byte_off)
{
  return *reinterpret_cast<const f4*>(lds + byte_off);
}
__device__ __forceinline__ void lds_st4(char* lds, unsigned byte_off, const f4& v)
{
  *reinterpret_cast<f4*>(lds + byte_off) = v;
}
} // namespace mf

// Ring append: 16 B at (wave-uniform base + per-lane byte offset). WT = write-through (sc0 sc1): a launch that
// covers only a block or two would otherwise leave every ring line dirty in L2 and pay for the write-back
// when the kernel retires (measured 3 us of a 28 us launch at 256 streams); long launches keep the default
// write-back policy, which is ~3% faster in steady state.
template <bool WT>
__device__ __forceinline__ void ring_store(char* base, unsigned off, mf::f4 v)
{
  if constexpr (WT)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v),
                                           __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000),
                                           (int)off, 0, /*sc0 sc1*/ 17);
  else
    *reinterpret_cast<mf::f4*>(base + off) = v;
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
  // position s0 + lane of this launch; the stream it stands for comes from the optional map (a batch whose
  // streams run different submodels launches each group over its member list)
  const bool live = s0 + lane < a.n_streams;
  const int stream = live ? (a.stream_map ? a.stream_map[s0 + lane] : s0 + lane) : 0;
  const int H = a.hidden, NL = a.n_layers, I0 = a.input_size;
  const int in_ch = a.in_ch, out_ch = a.out_ch;
  float* tile_in = lds; // [in_ch][64][65]
  float* tile_out = tile_in + in_ch * kBlock * 65; // [out_ch][64][65]
  float* hs = tile_out + out_ch * kBlock * 65; // [NL][H][64]
  float* cs = hs + NL * H * kBlock; // [NL][H][64]
  float* ifgo = cs + NL * H * kBlock; // [4H][64]

  // load recurrent state
  float* st = a.state + (size_t)stream * a.state_stride;
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
        const int s = __shfl(stream, r); // the stream of position s0 + r
        float v = 0.0f;
        if (a.in && s0 + r < a.n_streams && lane < nvalid)
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
          const int s = __shfl(stream, r);
          if (s0 + r < a.n_streams && lane < nvalid)
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
// LSTM on the matrix cores: 16 streams per wavefront, the per-sample [4H x (I + H)] . [x; h] of all 16 streams
// is a handful of v_mfma_f32_16x16x4_f32 (columns = streams). Weight rows are permuted (plan.h) so that lane
// group u of a unit tile receives the i, f, g, o pre-activations of ONE hidden unit: the gate math and the c / h
// update stay in that lane, and the lane's new h is exactly the B operand it feeds into the next time step /
// next layer. Everything (tiles, h, c, I/O tiles) lives in LDS; one wavefront per workgroup, no barriers.
// lstm.nam (H = 3): 2 MFMAs + one head MFMA per sample instead of 60 scalar-weight FMAs with exposed latency.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void nam_lstm_mfma_kernel(const float* __restrict__ blob, const LSTMArgs a)
{
  using mf::f4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  const int grp = lane >> 4, j = lane & 15;
  const int s0 = blockIdx.x * 16;
  const bool live = s0 + j < a.n_streams;
  const int stream = live ? (a.stream_map ? a.stream_map[s0 + j] : s0 + j) : 0;
  const int H = a.hidden, NL = a.n_layers, I0 = a.input_size, NT = a.mf_nt;
  const int in_ch = a.in_ch, out_ch = a.out_ch;
  const int KI0 = (I0 + 3) / 4;
  float* region = lds; // tiles, biases (plan.h)
  float* hbuf = region + a.mf_floats; // [2][NL][4 NT][16]
  float* cbuf = hbuf + 2 * NL * 4 * NT * 16; // [NL][4 NT][16]
  float* xin = cbuf + NL * 4 * NT * 16; // [in_ch][16][65]
  float* yout = xin + in_ch * 16 * 65; // [out_ch][16][65]
  const int hstride = NL * 4 * NT * 16; // one time parity of hbuf

  for (int i = lane; i < a.mf_floats; i += 64)
    region[i] = blob[a.mf_off + i];
  // recurrent state: lane (grp, j) owns unit 4T + grp of stream j
  float* st = a.state + (size_t)stream * a.state_stride;
  for (int l = 0; l < NL; l++)
    for (int T = 0; T < NT; T++)
    {
      const int u = 4 * T + grp;
      const bool ok = live && u < H;
      hbuf[hstride + (l * 4 * NT + u) * 16 + j] = ok ? st[(l * 2 + 0) * H + u] : 0.0f; // parity 1 = "time -1"
      cbuf[(l * 4 * NT + u) * 16 + j] = ok ? st[(l * 2 + 1) * H + u] : 0.0f;
    }
  int par = 0; // parity of the time step being computed
  for (int f0 = 0; f0 < a.n_frames; f0 += kBlock)
  {
    const int nvalid = min(kBlock, a.n_frames - f0);
    // coalesced input tile: row r = stream of position s0 + r, lane = frame
    for (int c = 0; c < in_ch; c++)
      for (int r = 0; r < 16; r++)
      {
        const int s = __shfl(stream, r);
        float v = 0.0f;
        if (a.in && s0 + r < a.n_streams && lane < nvalid)
          v = a.in[((size_t)s * in_ch + c) * a.io_stride + f0 + lane];
        xin[(c * 16 + r) * 65 + lane] = v;
      }
    for (int t = 0; t < nvalid; t++)
    {
      const float* hprev = hbuf + (par ^ 1) * hstride; // h(t - 1)
      float* hcur = hbuf + par * hstride; // h(t)
      for (int l = 0; l < NL; l++)
      {
        const int KI = l == 0 ? KI0 : NT;
        const float* tiles = region + a.mf_layer_tiles[l];
        const float* bias = region + a.mf_layer_bias[l];
        for (int T = 0; T < NT; T++)
        {
          f4 acc = *reinterpret_cast<const f4*>(bias + (T * 4 + grp) * 4); // i, f, g, o biases of unit 4T + grp
          const float* tl = tiles + (size_t)T * (KI + NT) * 64 + lane;
          for (int s = 0; s < KI; s++) // layer input: x(t) or the layer below's h(t)
          {
            const int e = 4 * s + grp;
            const float b = l == 0 ? (e < I0 ? xin[(e * 16 + j) * 65 + t] : 0.0f) : hcur[((l - 1) * 4 * NT + e) * 16 + j];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(tl[s * 64], b, acc, 0, 0, 0);
          }
          for (int s = 0; s < NT; s++) // this layer's h(t - 1)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(tl[(KI + s) * 64], hprev[(l * 4 * NT + 4 * s + grp) * 16 + j], acc, 0, 0, 0);
          const int u = 4 * T + grp;
          const float cprev = cbuf[(l * 4 * NT + u) * 16 + j];
          float cn, hn;
          if (a.fast)
          {
            cn = mf::fast_sigmoid_hw(acc[1]) * cprev + mf::fast_sigmoid_hw(acc[0]) * mf::fast_tanh_hw(acc[2]);
            hn = mf::fast_sigmoid_hw(acc[3]) * mf::fast_tanh_hw(cn);
          }
          else
          {
            cn = mf::sigmoid_hw(acc[1]) * cprev + mf::sigmoid_hw(acc[0]) * mf::tanh_hw(acc[2]);
            hn = mf::sigmoid_hw(acc[3]) * mf::tanh_hw(cn);
          }
          cbuf[(l * 4 * NT + u) * 16 + j] = cn;
          hcur[(l * 4 * NT + u) * 16 + j] = hn;
        }
      }
      // head: y = Wh . h_top(t) + bh; lane group g receives output channels 4g..4g+3
      {
        f4 acc = *reinterpret_cast<const f4*>(region + a.mf_head_bias + grp * 4);
        const float* tl = region + a.mf_head_tiles + lane;
        for (int s = 0; s < NT; s++)
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(tl[s * 64], hcur[((NL - 1) * 4 * NT + 4 * s + grp) * 16 + j], acc, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; e++)
          if (4 * grp + e < out_ch)
            yout[((4 * grp + e) * 16 + j) * 65 + t] = acc[e];
      }
      par ^= 1;
    }
    if (a.out)
      for (int c = 0; c < out_ch; c++)
        for (int r = 0; r < 16; r++)
        {
          const int s = __shfl(stream, r);
          if (s0 + r < a.n_streams && lane < nvalid)
            a.out[((size_t)s * out_ch + c) * a.io_stride + f0 + lane] = yout[(c * 16 + r) * 65 + lane];
        }
  }
  // the last computed step has parity par ^ 1
  const float* hlast = hbuf + (par ^ 1) * hstride;
  for (int l = 0; l < NL; l++)
    for (int T = 0; T < NT; T++)
    {
      const int u = 4 * T + grp;
      if (live && u < H)
      {
        st[(l * 2 + 0) * H + u] = hlast[(l * 4 * NT + u) * 16 + j];
        st[(l * 2 + 1) * H + u] = cbuf[(l * 4 * NT + u) * 16 + j];
      }
    }
}

// The same kernel for small models (<= 2 layers, <= 24 hidden units, <= 4 inputs), fully unrolled: every A
// tile value, bias, h and c of the lane stays in registers for the whole launch; per sample only the input is
// read from LDS and the output written to it. lstm.nam: ~0.2 us per sample step instead of ~0.9.
template <int NL, int NT>
__global__ __launch_bounds__(64) void nam_lstm_mfma_reg_kernel(const float* __restrict__ blob, const LSTMArgs a)
{
  using mf::f4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  const int grp = lane >> 4, j = lane & 15;
  const int s0 = blockIdx.x * 16;
  const bool live = s0 + j < a.n_streams;
  const int stream = live ? (a.stream_map ? a.stream_map[s0 + j] : s0 + j) : 0;
  const int H = a.hidden, I0 = a.input_size;
  const int in_ch = a.in_ch, out_ch = a.out_ch;
  float* xin = lds; // [in_ch][16][65]
  float* yout = xin + in_ch * 16 * 65; // [out_ch][16][65]
  const float* region = blob + a.mf_off;

  float wi[NL][NT][NT], wr[NL][NT][NT], h[NL][NT], c[NL][NT], wh[NT];
  f4 bias[NL][NT];
#pragma unroll
  for (int l = 0; l < NL; l++)
  {
    const int KI = l == 0 ? 1 : NT; // input_size <= 4: one k-step
    const float* tiles = region + a.mf_layer_tiles[l];
#pragma unroll
    for (int T = 0; T < NT; T++)
    {
#pragma unroll
      for (int s = 0; s < NT; s++)
      {
        wi[l][T][s] = s < KI ? tiles[(T * (KI + NT) + s) * 64 + lane] : 0.0f;
        wr[l][T][s] = tiles[(T * (KI + NT) + KI + s) * 64 + lane];
      }
      bias[l][T] = *reinterpret_cast<const f4*>(region + a.mf_layer_bias[l] + (T * 4 + grp) * 4);
    }
  }
#pragma unroll
  for (int s = 0; s < NT; s++)
    wh[s] = region[a.mf_head_tiles + s * 64 + lane];
  const f4 hbias = *reinterpret_cast<const f4*>(region + a.mf_head_bias + grp * 4);
  float* st = a.state + (size_t)stream * a.state_stride;
#pragma unroll
  for (int l = 0; l < NL; l++)
#pragma unroll
    for (int T = 0; T < NT; T++)
    {
      const int u = 4 * T + grp;
      const bool ok = live && u < H;
      h[l][T] = ok ? st[(l * 2 + 0) * H + u] : 0.0f;
      c[l][T] = ok ? st[(l * 2 + 1) * H + u] : 0.0f;
    }
  const bool fast = a.fast != 0;
  for (int f0 = 0; f0 < a.n_frames; f0 += kBlock)
  {
    const int nvalid = min(kBlock, a.n_frames - f0);
    for (int ch = 0; ch < in_ch; ch++)
      for (int r = 0; r < 16; r++)
      {
        const int s = __shfl(stream, r);
        float v = 0.0f;
        if (a.in && s0 + r < a.n_streams && lane < nvalid)
          v = a.in[((size_t)s * in_ch + ch) * a.io_stride + f0 + lane];
        xin[(ch * 16 + r) * 65 + lane] = v;
      }
    const int xrow = (min(grp, in_ch - 1) * 16 + j) * 65; // lane group g feeds input element g (zero weights beyond I0)
    float xv = xin[xrow];
    float x1 = xin[xrow + 1]; // the inputs are read two steps ahead: an LDS round trip per step is not on the chain
    // where this lane's four head outputs go (rows 4 grp + e of the output tile); lanes without a row write to a pad
    int yrow[4];
#pragma unroll
    for (int e = 0; e < 4; e++)
      yrow[e] = 4 * grp + e < out_ch ? ((4 * grp + e) * 16 + j) * 65 : out_ch * 16 * 65 + lane;
    const int n_store = min(4, out_ch); // rows 0..n_store-1 exist in lane group 0 (uniform: skips whole stores)
    // Off the recurrence's critical path: the input half of layer 0 (bias + Wi . x_t) is issued one step ahead, and a
    // step's output is stored one step later (the store would otherwise sit, in order, behind the head MFMA's result)
    f4 pre0[NT];
#pragma unroll
    for (int T = 0; T < NT; T++)
      pre0[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(wi[0][T][0], grp < I0 ? xv : 0.0f, bias[0][T], 0, 0, 0);
    f4 ypend = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < nvalid; t++)
    {
      const float xnext = x1; // next step's input
      x1 = xin[xrow + min(t + 2, kBlock - 1)];
      float hn[NL][NT];
#pragma unroll
      for (int l = 0; l < NL; l++)
      {
        // the NT unit tiles of a layer are independent: their MFMA chains are issued interleaved (k-step outer,
        // tile inner) so no MFMA waits on its predecessor, then the gate math of all tiles follows
        f4 acc[NT];
#pragma unroll
        for (int T = 0; T < NT; T++)
          acc[T] = l == 0 ? pre0[T] : bias[l][T];
        if (l > 0)
        {
#pragma unroll
          for (int s = 0; s < NT; s++)
#pragma unroll
            for (int T = 0; T < NT; T++)
              acc[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(wi[l][T][s], hn[l > 0 ? l - 1 : 0][s], acc[T], 0, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < NT; s++)
#pragma unroll
          for (int T = 0; T < NT; T++)
            acc[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[l][T][s], h[l][s], acc[T], 0, 0, 0);
        if (l == 0)
        {
          // next step's input half and the previous step's output store, in the shadow of the MFMAs above
          const float xn = grp < I0 ? xnext : 0.0f;
#pragma unroll
          for (int T = 0; T < NT; T++)
            pre0[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(wi[0][T][0], xn, bias[0][T], 0, 0, 0);
          if (t > 0)
          {
#pragma unroll
            for (int e = 0; e < 4; e++)
              if (e < n_store)
                yout[yrow[e] + t - 1] = ypend[e];
          }
        }
#pragma unroll
        for (int T = 0; T < NT; T++)
        {
          float cn, hv;
          if (fast)
          {
            cn = mf::fast_sigmoid_hw(acc[T][1]) * c[l][T] + mf::fast_sigmoid_hw(acc[T][0]) * mf::fast_tanh_hw(acc[T][2]);
            hv = mf::fast_sigmoid_hw(acc[T][3]) * mf::fast_tanh_hw(cn);
          }
          else
          {
            cn = mf::sigmoid_hw(acc[T][1]) * c[l][T] + mf::sigmoid_hw(acc[T][0]) * mf::tanh_hw(acc[T][2]);
            hv = mf::sigmoid_hw(acc[T][3]) * mf::tanh_hw(cn);
          }
          c[l][T] = cn;
          hn[l][T] = hv;
        }
      }
      ypend = hbias;
#pragma unroll
      for (int s = 0; s < NT; s++)
        ypend = __builtin_amdgcn_mfma_f32_16x16x4f32(wh[s], hn[NL - 1][s], ypend, 0, 0, 0);
#pragma unroll
      for (int l = 0; l < NL; l++)
#pragma unroll
        for (int T = 0; T < NT; T++)
          h[l][T] = hn[l][T];
      xv = xnext;
    }
    if (nvalid > 0)
    {
#pragma unroll
      for (int e = 0; e < 4; e++)
        if (e < n_store)
          yout[yrow[e] + nvalid - 1] = ypend[e];
    }
    if (a.out)
      for (int ch = 0; ch < out_ch; ch++)
        for (int r = 0; r < 16; r++)
        {
          const int s = __shfl(stream, r);
          if (s0 + r < a.n_streams && lane < nvalid)
            a.out[((size_t)s * out_ch + ch) * a.io_stride + f0 + lane] = yout[(ch * 16 + r) * 65 + lane];
        }
  }
#pragma unroll
  for (int l = 0; l < NL; l++)
#pragma unroll
    for (int T = 0; T < NT; T++)
    {
      const int u = 4 * T + grp;
      if (live && u < H)
      {
        st[(l * 2 + 0) * H + u] = h[l][T];
        st[(l * 2 + 1) * H + u] = c[l][T];
      }
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
  const bool wlds = a.blob_floats > 0;
  if (lds_bytes > 64 * 1024)
  {
    hipError_t e = hipFuncSetAttribute(wlds ? reinterpret_cast<const void*>(nam_generic_kernel<true>)
                                            : reinterpret_cast<const void*>(nam_generic_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess)
      return e;
  }
  if (wlds)
    hipLaunchKernelGGL(nam_generic_kernel<true>, dim3(n_blocks), dim3(64), lds_bytes, stream, a.ops, a.blob, a);
  else
    hipLaunchKernelGGL(nam_generic_kernel<false>, dim3(n_blocks), dim3(64), lds_bytes, stream, a.ops, a.blob, a);
  return hipGetLastError();
}

hipError_t launch_a1(const A1Args& a, int n_blocks, hipStream_t stream)
{
  hipLaunchKernelGGL(nam_a1_kernel, dim3(n_blocks), dim3(64), 0, stream, a.plan, a.blob, a);
  return hipGetLastError();
}

// ================================================================================================
// nam_a1_mfma_kernel — wave-specialised fp32-MFMA kernel: 8 wavefronts per stream, one job per LAYER.
//   waves 0-3 (compute): per job one barrier, 3 LDS reads on the critical path (two shifted taps + the input
//                        sample), 16-20 MFMAs, activation, publish x. The job's weight tiles and constants are
//                        already in registers: they are read from LDS one job ahead, in the shadow of the MFMAs.
//                        No vector-memory instructions except the output store.
//   waves 4-7 (movers) : per job drop the successor's prefetched history (3 x 16 B per lane) and the weight
//                        tiles of the job after it (16 B per lane) into the LDS double buffers, append the
//                        job's input rows (LDS window) to its HBM ring, and issue the loads of the job
//                        D + 1 ahead (D = the plan's ws_prefetch). At block boundaries they also materialise
//                        x0 = rechannel * input and the input samples in LDS.
// Each SIMD hosts one compute and one mover wave, so the mover's address arithmetic / memory instructions
// fill the issue slots the compute wave leaves between dependent MFMA / VALU instructions.
// Rechannel and head-rechannel steps ride on the neighbouring layer jobs (plan.h: CDesc / VDesc).
// ================================================================================================
namespace ws
{
using mf::f4;
constexpr int SC = kMfSC;
struct HSlot
{
  f4 h[2]; // the job's two history sets (plan.h, VDesc)
  f4 tile; // 16 B of the weight tiles of the job AFTER the one the history belongs to
  float inp; // input sample of frame hfr of the block the job belongs to
};
struct Ops // one job's register-resident operands (compute waves)
{
  f4 t[4]; // A tiles: conv tap 0,1,2 | layer1x1
  f4 xt; // extra tile (rechannel / head rechannel) when the job has one
  f4 bv, mv, b1v, ev; // conv bias | input mixin | 1x1 bias | extra (rechannel column or head bias)
};
} // namespace ws

template <int ACT_T, bool WT, bool PROF, int D>
__global__ __launch_bounds__(512) void nam_a1_mfma_kernel(const A1Plan* __restrict__ P, const float* __restrict__ blob,
                                                        const A1Args a)
{
  using namespace mf;
  using ws::HSlot;
  using ws::Ops;
  constexpr int SC = ws::SC;
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  char* const lds = reinterpret_cast<char*>(lds_f);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = uni(tid >> 6);
  const int stream = a.stream_map ? a.stream_map[blockIdx.x] : (int)blockIdx.x;
  float* st = a.state + (size_t)stream * a.state_stride;
  char* stb = reinterpret_cast<char*>(st);
  const int NJ = a.n_mjobs;
  const int n_blocks = (a.n_frames + kBlock - 1) / kBlock;
  const int total = n_blocks * NJ;
  constexpr int kUnroll = (D % 2 == 0) ? D : 2 * D; // both roles run a multiple of this many jobs (= barriers)
  const int total_pad = (total + kUnroll - 1) / kUnroll * kUnroll;
  const unsigned lds_tiles_b = (unsigned)a.lds_tiles_b, lds_cond_b = (unsigned)a.lds_cond_b;

  // constants table and extra tiles -> LDS (all 512 threads; visible at the prologue barrier)
  {
    const float* __restrict__ csrc = blob + a.consts_off;
    const float* __restrict__ xsrc = blob + a.xt_off;
    constexpr int NC = kWsJobMax * 64 / 512, NX = kWsXtMax * 256 / 512;
    float cv[NC], xv[NX];
#pragma unroll
    for (int i = 0; i < NC; i++)
      cv[i] = csrc[tid + 512 * i]; // the blob tables are padded to their maximum sizes
#pragma unroll
    for (int i = 0; i < NX; i++)
      xv[i] = xsrc[tid + 512 * i];
#pragma unroll
    for (int i = 0; i < NC; i++)
      if (tid + 512 * i < NJ * 64)
        lds_f[kWsConstsOff + tid + 512 * i] = cv[i];
#pragma unroll
    for (int i = 0; i < NX; i++)
      if (tid + 512 * i < a.n_xt * 256)
        lds_f[a.lds_xt_b / 4 + tid + 512 * i] = xv[i];
  }

  long long bar_cycles = 0, t_begin = 0; // PROF: cycles this wave spent inside barriers / total
  long long seg[5] = {0, 0, 0, 0, 0}; // PROF (compute waves): taps ready | conv done | x updated | published | job end
  long long t_seg = 0;
#define NAM_WS_STAMP(k, ...) \
  if constexpr (PROF) \
  { \
    asm volatile("" ::__VA_ARGS__); \
    const long long t_now = __builtin_readcyclecounter(); \
    seg[k] += t_now - t_seg; \
    t_seg = t_now; \
  }
  if constexpr (PROF)
    t_begin = __builtin_readcyclecounter();
  auto job_barrier = [&]() {
    if constexpr (PROF)
    {
      const long long t0 = __builtin_readcyclecounter();
      lds_barrier();
      bar_cycles += __builtin_readcyclecounter() - t0;
    }
    else
      lds_barrier();
  };
  if (w < 4)
  {
    // ------------------------------------------------ compute role ------------------------------------
    const int g = lane >> 4; // channel quad: this lane owns channels 4g..4g+3
    const int frame = 16 * w + (lane & 15);
    float* out = a.out ? a.out + (size_t)stream * a.io_stride : nullptr;
    const float head_scale = a.head_scale;
    const float act_p0 = a.act_p0;
    const unsigned v_g16 = (unsigned)g * 16u;
    const unsigned v_tap = (unsigned)(frame * SC) * 4u;
    const unsigned v_cond = lds_cond_b + (unsigned)frame * 4u;
    const unsigned v_lane16 = (unsigned)lane * 16u;
    const unsigned v_gh8 = (unsigned)(g & 1) * 16u + (unsigned)(g >> 1) * 8u; // half layout: the lane's channel pair
    // a job's operands: 4 tiles from tile buffer `tbuf`, its extra tile, 4 constant vectors
    auto load_ops = [&](Ops& o, const CDesc& J, int tbuf) {
      const unsigned a_t = v_lane16 + lds_tiles_b + (unsigned)tbuf * (kWsTileFloats * 4u);
#pragma unroll
      for (int q = 0; q < 4; q++)
        o.t[q] = lds_ld4(lds, a_t + 1024u * q);
      o.xt = lds_ld4(lds, v_lane16 + (unsigned)J.xt_b);
      const unsigned a_c = v_g16 + (unsigned)J.consts_b;
      o.bv = lds_ld4(lds, a_c);
      o.mv = lds_ld4(lds, a_c + 64u);
      o.b1v = lds_ld4(lds, a_c + 128u);
      o.ev = lds_ld4(lds, a_c + 192u);
    };
    f4 x = {0.f, 0.f, 0.f, 0.f}, head = {0.f, 0.f, 0.f, 0.f};
    int ji = 0, blk = 0;
    int nvalid = min(kBlock, a.n_frames);
    // descriptors: J (this job) <- Dn (next job: its operands are prefetched during this job) <- Dnn (the job after,
    // scalar-loaded in the shadow of this job's MFMAs so that no barrier's lgkmcnt(0) ever waits for it)
    CDesc Dn = P->cdesc[0];
    CDesc Dnn = P->cdesc[1];
    Ops ops[2];
    job_barrier(); // prologue barrier: consts, extra tiles, job 0's tiles / history / x0 are in LDS
    load_ops(ops[0], Dn, 0);

    for (int q0 = 0; q0 < total_pad; q0 += 2)
    {
#pragma unroll
      for (int u = 0; u < 2; u++)
      {
        const bool active = q0 + u < total;
        const CDesc J = Dn;
        Dn = Dnn;
        const int flags_rt = active ? J.flags : 0;
        const Ops& O = ops[u];
        job_barrier();
        if constexpr (PROF)
          t_seg = __builtin_readcyclecounter();
        const float cond = *reinterpret_cast<const float*>(lds + (v_cond + (unsigned)(blk & 1) * (kBlock * 4u)));
        // everything between the barrier and the publish, for NK k-steps per matrix (4: full layout, 2: half)
        // PLAIN: an ordinary layer (no array entry / exit work): the flag tests below fold away at compile time
        auto job_body = [&](auto nk_tag, auto plain_tag) {
          constexpr int NK = decltype(nk_tag)::value;
          constexpr bool PLAIN = decltype(plain_tag)::value;
          const int flags = PLAIN ? (int)CD_LAYER : flags_rt;
          // critical-path operand reads: the two shifted taps. Full layout: the lane's channel quad (16 B);
          // half layout: the two channels this lane feeds to the MFMAs (8 B).
          f4 bt0, bt1;
          if constexpr (NK == 4)
          {
            const unsigned a_tap = v_tap + min(v_g16, (unsigned)(J.gp & 0xff));
            bt0 = lds_ld4(lds, a_tap + (unsigned)J.tap0_b);
            bt1 = lds_ld4(lds, a_tap + (unsigned)J.tap1_b);
          }
          else
          {
            const f2 p0 = *reinterpret_cast<const f2*>(lds + (v_tap + v_gh8 + (unsigned)J.tap0_b));
            const f2 p1 = *reinterpret_cast<const f2*>(lds + (v_tap + v_gh8 + (unsigned)J.tap1_b));
            bt0 = f4{p0[0], p0[1], 0.f, 0.f};
            bt1 = f4{p1[0], p1[1], 0.f, 0.f};
          }
          NAM_WS_STAMP(0, "v"(bt0), "v"(bt1), "v"(cond))
          if (flags & CD_X0)
          {
            x = O.ev * cond; // ev = first array's rechannel column (in_size == 1)
            head = f4{0.f, 0.f, 0.f, 0.f};
          }
          else if (flags & CD_PRE_HEAD) // previous array's head rechannel + bias, in this array's layout
            head = ((flags & CD_PREV_HALF) ? mfma_n<2>(O.xt, head, f4{0.f, 0.f, 0.f, 0.f})
                                           : mfma_n<4>(O.xt, head, f4{0.f, 0.f, 0.f, 0.f}))
                   + O.ev;
          // dilated conv: 3 taps x NK k-steps. Tap 2 (the current frame) multiplies the lane's own x and needs
          // nothing from LDS, so its chain goes first and covers the latency of the two shifted-tap reads; the
          // conv bias and the input mixin ride in as initial accumulators. The next job's operands (its tiles
          // were dropped one job ago) are requested behind the taps, in the shadow of the MFMAs.
          load_ops(ops[u ^ 1], Dn, u ^ 1);
          {
            int jn = ji + 2;
            if (jn >= NJ)
              jn -= NJ;
            Dnn = P->cdesc[jn];
          }
          f4 acc0 = O.mv * cond, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = O.bv;
#pragma unroll
          for (int s = 0; s < NK; s++)
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(O.t[2][s], x[s], acc2, 0, 0, 0);
#pragma unroll
          for (int s = 0; s < NK; s++)
          {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(O.t[0][s], bt0[s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(O.t[1][s], bt1[s], acc1, 0, 0, 0);
          }
          const f4 pre = (acc0 + acc1) + acc2;
          NAM_WS_STAMP(1, "v"(pre))
          if (flags & CD_LAYER)
          {
            // half layout: only elements 0, 1 of a lane ever feed an MFMA (z into the 1x1, head into the head
            // rechannel), elements 2, 3 are the partner lane group's copies: two activations per lane, not four
            const f4 z = act4<ACT_T>(J.act, NK == 2 ? f4{pre[0], pre[1], pre[0], pre[1]} : pre, act_p0);
            head += z;
            // layer1x1 as two chains; the residual and the 1x1 bias are the initial accumulator
            f4 y0 = x + O.b1v, y1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NK; s += 2)
            {
              y0 = __builtin_amdgcn_mfma_f32_16x16x4f32(O.t[3][s], z[s], y0, 0, 0, 0);
              y1 = __builtin_amdgcn_mfma_f32_16x16x4f32(O.t[3][s + 1], z[s + 1], y1, 0, 0, 0);
            }
            x = y0 + y1;
            NAM_WS_STAMP(2, "v"(x))
            if (flags & CD_POST_OUT)
            {
              const f4 hout = mfma_n<NK>(O.xt, head, f4{0.f, 0.f, 0.f, 0.f}) + O.ev;
              if (out && g == 0 && frame < nvalid)
                out[(size_t)blk * kBlock + frame] = head_scale * hout[0];
            }
            else
            {
              if (flags & CD_POST_RECH)
                x = mfma_n<NK>(O.xt, x, f4{0.f, 0.f, 0.f, 0.f}); // next array's rechannel (no bias), its layout
              if (v_g16 <= (unsigned)(J.gp >> 8))
                lds_st4(lds, v_tap + v_g16 + (unsigned)J.pub_b, x);
            }
          }
          // What the job barrier waits for anyway, stated at the end of every variant: the four variants are
          // laid out one after the other behind flag tests, so without it the waitcnt pass carries one variant's
          // outstanding operand reads into the entry of the next and puts an lgkmcnt(0) in front of its first MFMA.
          // (the per-variant asm comment keeps the optimiser from sinking the four waits into one at the join.)
          __builtin_amdgcn_s_waitcnt(0xc07f);
          asm volatile("; end of job body nk=%0 plain=%1" ::"n"(NK), "n"((int)PLAIN));
        };
        const bool plain = (flags_rt & ~CD_HALF) == CD_LAYER;
        if (flags_rt & CD_HALF)
        {
          if (plain)
            job_body(std::integral_constant<int, 2>{}, std::true_type{});
          else
            job_body(std::integral_constant<int, 2>{}, std::false_type{});
        }
        else
        {
          if (plain)
            job_body(std::integral_constant<int, 4>{}, std::true_type{});
          else
            job_body(std::integral_constant<int, 4>{}, std::false_type{});
        }
        NAM_WS_STAMP(3, "v"(x))
        if (active && ++ji == NJ)
        {
          ji = 0;
          blk++;
          nvalid = min(kBlock, a.n_frames - blk * kBlock);
        }
        NAM_WS_STAMP(4, "s"(ji))
      }
    }
  }
  else
  {
    // ------------------------------------------------ mover role --------------------------------------
    const int mtid = tid - 256;
    const int hfr = 16 * (w - 4) + (lane >> 2); // frame inside a 64-frame set
    const unsigned v_hq16 = (unsigned)(lane & 3) * 16u; // channel quad
    const unsigned v_hist = (unsigned)(hfr * SC + 4 * (lane & 3)) * 4u;
    const unsigned v_mt16 = (unsigned)mtid * 16u;
    int* wpos_tbl = reinterpret_cast<int*>(st);
    const float* in = a.in ? a.in + (size_t)stream * a.io_stride : nullptr;
    const char* ibase = in ? reinterpret_cast<const char*>(in) : stb; // silence: any valid word, masked later
    const char* tiles0 = reinterpret_cast<const char*>(blob + a.tiles_off);
    int wposv = wpos_tbl[lane]; // lane r = write position of ring r
    const int ring_len_v = P->ring_len_by_id[lane];
    const f4 r1q = *reinterpret_cast<const f4*>(blob + a.r1_off + 4 * (lane & 3));

    // history of one job + the tiles of job `tjob`
    auto fetch = [&](HSlot& s, int f_rbase, int f_R, int f_LA, int f_LB, int f_ring_id, int f_q16max, bool next_block,
                     int jblk, int tjob) {
      int wp = __builtin_amdgcn_readlane(wposv, f_ring_id);
      if (next_block)
      {
        wp += kBlock;
        if (wp >= f_R)
          wp -= f_R;
      }
      const unsigned vq = min(v_hq16, (unsigned)f_q16max) + (unsigned)f_rbase;
      const unsigned cmul = (unsigned)f_q16max + 16u;
      const int Ls[2] = {f_LA, f_LB};
#pragma unroll
      for (int t = 0; t < 2; t++)
      {
        int sb = wp - Ls[t];
        if (sb < 0)
          sb += f_R;
        const unsigned v = (unsigned)(hfr + sb);
        const unsigned idx = min(v, v - (unsigned)f_R);
        // a job without a second set still issues the load (the number of loads in flight stays static),
        // but every lane reads the same cached 16 bytes
        const unsigned off = (t == 1 && f_LB == 0) ? 0u : __umul24(idx, cmul) + vq;
        s.h[t] = *reinterpret_cast<const f4*>(stb + off);
      }
      s.tile = *reinterpret_cast<const f4*>(tiles0 + ((unsigned)tjob * (kWsTileFloats * 4u) + v_mt16));
      int fi = jblk * kBlock + hfr;
      fi = min(fi, a.n_frames - 1);
      s.inp = *reinterpret_cast<const float*>(ibase + (in ? (unsigned)fi * 4u : 0u));
    };
    // drop a job's history (+ the following job's tiles) into LDS; for a block's first job also x0 and the inputs
    auto drop = [&](const HSlot& s, const VDesc& J, int succ_blk, int tbuf) {
      lds_st4(lds, v_hist + (unsigned)J.st_a_b, s.h[0]);
      if (J.flags & MV_SUCC_B)
        lds_st4(lds, v_hist + (unsigned)J.st_b_b, s.h[1]);
      lds_st4(lds, v_mt16 + lds_tiles_b + (unsigned)tbuf * (kWsTileFloats * 4u), s.tile);
      if (J.flags & MV_SUCC_FIRST)
      {
        const bool live = in && (succ_blk * kBlock + hfr < a.n_frames);
        const float iv = live ? s.inp : 0.0f;
        lds_st4(lds, v_hist + (unsigned)J.st_x0_b, r1q * iv);
        if ((lane & 3) == 0)
          *reinterpret_cast<float*>(lds + (lds_cond_b + (unsigned)((succ_blk & 1) * kBlock + hfr) * 4u)) = iv;
      }
    };

    HSlot slot[D];
    // job 0's tiles go straight to tile buffer 0; slot u = history of job u + tiles of job u + 1
    const f4 tile0 = *reinterpret_cast<const f4*>(tiles0 + v_mt16);
#pragma unroll
    for (int u = 0; u < D; u++)
    {
      const VDesc F = P->vdesc[u + NJ - 1 - D]; // the descriptor whose f_* fields describe job u
      fetch(slot[u], F.f_rbase, F.f_R, F.f_LA, F.f_LB, F.f_ring_id, F.f_q16max, false, 0, u + 1);
    }
    int ji = 0, blk = 0;
    int fj = D + 1, fblk = 0; // job / block whose history is fetched next
    int ftile = D + 2; // job whose tiles are fetched next (NJ >= D + 3)
    int nvalid = min(kBlock, a.n_frames);
    {
      // "job -1": job 0's history (and x0 / inputs of block 0) go to LDS, slot 0 is refilled with job D
      const VDesc J = P->vdesc[NJ - 1];
      lds_st4(lds, v_mt16 + lds_tiles_b, tile0);
      drop(slot[0], J, 0, 1);
      fetch(slot[0], J.f_rbase, J.f_R, J.f_LA, J.f_LB, J.f_ring_id, J.f_q16max, false, 0, D + 1);
    }
    VDesc Dn = P->vdesc[0];
    job_barrier(); // prologue barrier (matches the compute role)
    for (int q0 = 0; q0 < total_pad; q0 += D)
    {
#pragma unroll
      for (int u = 0; u < D; u++)
      {
        const bool active = q0 + u < total;
        const VDesc J = Dn;
        const int flags = active ? J.flags : 0;
        const int un = (u + 1) % D;
        job_barrier();
        Dn = P->vdesc[ji + 1 == NJ ? 0 : ji + 1]; // after the barrier: its lgkmcnt(0) must not wait for this load
        // this job's input rows (published by the previous job / dropped as x0) -> history ring
        if ((flags & MV_RING) && hfr < nvalid && v_hq16 <= (unsigned)J.q16max)
        {
          const f4 xin = lds_ld4(lds, v_hist + (unsigned)J.ap_src_b);
          const unsigned v = (unsigned)(__builtin_amdgcn_readlane(wposv, J.ring_id) + hfr);
          const unsigned widx = min(v, v - (unsigned)J.R);
          ring_store<WT>(stb, __umul24(widx, (unsigned)J.q16max + 16u) + v_hq16 + (unsigned)J.ring_b, xin);
        }
        // successor's history and the tiles of the job after it -> LDS (the halves of the double buffers
        // nobody reads during this job), then refill the slot (the other order — refill first — measured 6 % slower)
        drop(slot[un], J, blk + 1, (q0 + u) & 1);
        {
          const bool valid = fblk < n_blocks;
          fetch(slot[un], valid ? J.f_rbase : 0, valid ? J.f_R : 64, valid ? J.f_LA : 64, valid ? J.f_LB : 0,
                valid ? J.f_ring_id : 0, valid ? J.f_q16max : 0, valid && (fblk > blk), valid ? fblk : blk, ftile);
          if (++fj == NJ)
          {
            fj = 0;
            fblk++;
          }
          if (++ftile == NJ)
            ftile = 0;
        }
        if (active && ++ji == NJ)
        {
          ji = 0;
          wposv += nvalid;
          if (wposv >= ring_len_v)
            wposv -= ring_len_v;
          blk++;
          nvalid = min(kBlock, a.n_frames - blk * kBlock);
        }
      }
    }
    if (w == 4 && lane < a.n_rings)
      wpos_tbl[lane] = wposv;
  }
  if constexpr (PROF)
  {
    // rows 0..7 of the debug buffer: per wave of workgroup 0: {barrier cycles, total cycles}
    if (a.dbg && blockIdx.x == 0 && lane == 0)
    {
      a.dbg[w * 8 + 0] = bar_cycles;
      a.dbg[w * 8 + 1] = __builtin_readcyclecounter() - t_begin;
      for (int k = 0; k < 5; k++)
        a.dbg[w * 8 + 2 + k] = seg[k];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K-tap MFMA kernel (A2 family: single layer array, per-layer kernel sizes up to 16, head rechannel with taps)
// ------------------------------------------------------------------------------------------------
// One workgroup of four wavefronts per stream, wave w owns frames [16w, 16w + 16) of the 64-frame block, lane
// layout and MFMA operand mapping exactly as in nam_a1_mfma_kernel (full layout, or half layout for C = 8).
// A layer is a run of CHUNKS of up to kKtTaps taps (plan.h: KtDesc). Every tap's B operand is the lane's slice of
// frame (t - L): from the LDS copy of the layer input when that frame is inside the block, else from the layer's
// history ring in HBM — requested D CHUNKS AHEAD with buffer loads whose offset is out of range for lanes that do
// not need history (no memory traffic), together with that chunk's tap tiles. Only a layer's
// last chunk activates, runs the 1x1, publishes and meets the barrier: one barrier per layer. The head rechannel
// (A2: 16 taps over the head accumulator) is one more layer whose published input is the head accumulator.
// State layout, ring geometry and write positions are nam_a1_kernel's (the two are interchangeable mid-stream).
template <int NK, bool WT, int ACT_T>
__global__ __launch_bounds__(256) void nam_kt_mfma_kernel(const A1Plan* __restrict__ P, const float* __restrict__ blob,
                                                          const A1Args a)
{
  using mf::f2;
  using mf::f4;
  using fN = std::conditional_t<NK == 2, f2, f4>;
  using uN = std::conditional_t<NK == 2, __attribute__((ext_vector_type(2))) unsigned,
                                __attribute__((ext_vector_type(4))) unsigned>;
  // Everything a chunk needs from memory — history operands (a ring row written by an earlier launch comes from HBM:
  // 2-3 us), its tap tiles, the next block's input sample — is requested D chunks ahead, into one of D fixed
  // register sets, and consumed in request order: vmcnt retires in order, so a wait for anything younger would also
  // wait for every older request. The chunk loop is unrolled D times (one body per set); per chunk 1 store +
  // kOps loads, D * (kOps + 1) <= 63 outstanding.
  constexpr int D = NK == 2 ? 5 : 3;
  constexpr int NT = NK == 2 ? kKtTaps / 2 : kKtTaps; // 16-byte tile loads per chunk (half layout: two taps each)
  extern __shared__ __attribute__((aligned(16))) float lds_kt[];
  char* const lds = reinterpret_cast<char*>(lds_kt);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = uni(tid >> 6);
  const int g = lane >> 4;
  const int frame = 16 * w + (lane & 15);
  const int stream = a.stream_map ? a.stream_map[blockIdx.x] : (int)blockIdx.x;
  float* st = a.state + (size_t)stream * a.state_stride;
  const float* in = a.in ? a.in + (size_t)stream * a.io_stride : nullptr;
  float* out = a.out ? a.out + (size_t)stream * a.io_stride : nullptr;
  const int C = P->arr[0].channels;
  const int act = P->arr[0].act;
  const float act_p0 = a.act_p0, head_scale = a.head_scale;
  const int NCH = P->kt_chunks;
  const int n_blocks = (a.n_frames + kBlock - 1) / kBlock;
  const int total = n_blocks * NCH;
  const int total_pad = (total + D - 1) / D * D;
  const unsigned row_b = (unsigned)(C + 4) * 4u; // LDS row pitch of a published frame
  const unsigned buf_b = (kBlock + 1) * row_b; // two buffers, alternating per layer; row 0 of each is zero: a tap whose
                                               // frame lies before the block reads it and takes its operand from history
  const unsigned aux_b = 2u * buf_b; // LDS copy of the 1x1 tiles and the constants
  const unsigned ring_row_b = (unsigned)C * 4u;
  // the lane's B-operand slice of a frame row (its own channels, see nam_a1_mfma_kernel) and the quad it publishes
  const unsigned opnd_b = NK == 2 ? (unsigned)((g & 1) * 16 + (g >> 1) * 8) : min((unsigned)g * 16u, ring_row_b - 16u);
  const bool pub_lane = NK == 2 ? g < 2 : 4 * g < C;
  const unsigned quad_b = (unsigned)g * 16u;
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)st, 0, (int)(a.state_stride * 4), 0x00020000);
  // tiles: scalar base + scalar offset per load, the lane only contributes lane * 16; input samples: reads beyond the
  // launch's frames (or without an input) return 0
  const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)blob, 0, 0x7fffffff, 0x00020000);
  const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc((void*)(in ? in : st), 0, in ? a.n_frames * 4 : 0, 0x00020000);
  const int lane16 = lane * 16, frame4 = frame * 4;
  constexpr unsigned kOob = 0x7ffffff0u; // beyond num_records: the load returns 0 without touching memory
  int* wpos_tbl = reinterpret_cast<int*>(st);
  int wposv = lane < a.n_rings ? wpos_tbl[lane] : 0; // lane r = write position of ring r
  const int ring_len_v = P->ring_len_by_id[lane];
  const f4 rech = *reinterpret_cast<const f4*>(blob + P->kt_rech_off + g * 4);
  {
    const f4* __restrict__ src = reinterpret_cast<const f4*>(blob + P->kt_lds_src_off);
    const int n4 = P->kt_lds_floats / 4;
    for (int i = tid; i < n4; i += 256)
      *reinterpret_cast<f4*>(lds + aux_b + (unsigned)i * 16u) = src[i];
  }

  int blk = 0, ci = 0;
  struct Set
  {
    fN th[kKtTaps]; // history operands (lanes whose frame - L lies before the block)
    f4 tt[NT]; // tap tiles
    float cn; // input sample of the block after the chunk's
  };
  Set S[D];
  // operands of the chunk described by Dq, which lies `ahead` (0 / 1) blocks after the current one
  auto fetch = [&](Set& s, const KtDesc& Dq, int ahead) {
    int wp = __builtin_amdgcn_readlane(wposv, Dq.ring_id) + (ahead ? kBlock : 0);
    if (wp >= Dq.R)
      wp -= Dq.R;
#pragma unroll
    for (int i = 0; i < kKtTaps; i++)
    {
      // branch-free: lanes whose frame is inside the block (and taps the chunk does not have) get an offset beyond
      // the buffer; the rest row (wp + frame - L) mod R of the ring
      const int tl = frame - (i < Dq.ntaps ? Dq.L[i] : 0);
      int idx = wp + tl;
      idx += (idx >> 31) & Dq.R;
      const unsigned real = __umul24((unsigned)idx, ring_row_b) + ((unsigned)Dq.ring_b + opnd_b);
      const unsigned hmask = (unsigned)(tl >> 31); // all ones: history
      const unsigned off = (real & hmask) | (kOob & ~hmask);
      uN raw;
      if constexpr (NK == 2)
        raw = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)off, 0, 0);
      else
        raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0);
      s.th[i] = __builtin_bit_cast(fN, raw);
    }
#pragma unroll
    for (int i = 0; i < NT; i++)
      s.tt[i] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, lane16, Dq.tile_off * 4 + i * 1024, 0));
    s.cn = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_in, frame4, uni((blk + ahead + 1) * (kBlock * 4)), 0));
  };
  auto wrap = [&](int c) { return c >= NCH ? c - NCH : c; }; // NCH > D (plan.cpp)

  int nvalid = min(kBlock, a.n_frames);
  unsigned par = 0;
  float cond = (in && frame < nvalid) ? in[frame] : 0.0f;
  f4 x = rech * cond, head = {0.f, 0.f, 0.f, 0.f};
  f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  f4 pend = x; // rows published last (in LDS buffer `par`): appended to the consuming layer's ring by its first chunk
  if (tid < 64 && (unsigned)(tid & 31) * 4u < row_b)
    *reinterpret_cast<float*>(lds + ((unsigned)(tid >> 5) * buf_b + (unsigned)(tid & 31) * 4u)) = 0.0f;
  if (pub_lane)
    mf::lds_st4(lds, (unsigned)(frame + 1) * row_b + quad_b, x);
  // prologue: the operands of chunks 0 .. D-1
#pragma unroll
  for (int c = 0; c < D; c++)
  {
    const KtDesc Dq = P->kt_desc[c];
    fetch(S[c], Dq, 0);
  }
  KtDesc Dn = P->kt_desc[0];
  mf::lds_barrier();

  for (int q0 = 0; q0 < total_pad; q0 += D)
  {
#pragma unroll
    for (int u = 0; u < D; u++)
    {
      const bool active = q0 + u < total;
      const KtDesc J = Dn;
      const int flags = active ? J.flags : 0;
      const int ntaps = active ? J.ntaps : 0;
      Set& s = S[u];
      Dn = P->kt_desc[wrap(ci + 1)];
      const KtDesc Dq = P->kt_desc[wrap(ci + D)]; // the chunk this set is refilled for
      // in-block operands of this chunk's taps (rows of the layer input published before the last barrier), the
      // layer's constants and 1x1 tile
      fN lv[kKtTaps];
#pragma unroll
      for (int i = 0; i < kKtTaps; i++)
      {
        const int row = max(frame + 1 - J.L[i], 0);
        lv[i] = *reinterpret_cast<const fN*>(lds + (__umul24((unsigned)row, row_b) + (par * buf_b + opnd_b)));
      }
      const unsigned a_c = aux_b + (unsigned)J.consts_off + quad_b;
      const f4 bv = mf::lds_ld4(lds, a_c), mv = mf::lds_ld4(lds, a_c + 64u), b1v = mf::lds_ld4(lds, a_c + 128u);
      const fN w1 = *reinterpret_cast<const fN*>(lds + (aux_b + (unsigned)J.w1_off + (unsigned)lane * (NK * 4u)));
      // a layer's first chunk appends the layer input (still in `pend`) to the layer's ring
      {
        int v = __builtin_amdgcn_readlane(wposv, J.ring_id) + frame;
        if (v >= J.R)
          v -= J.R;
        const bool app = (flags & KT_FIRST) && (flags & KT_RING) && pub_lane && frame < nvalid;
        const unsigned off = app ? (unsigned)J.ring_b + (unsigned)v * ring_row_b + quad_b : kOob;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, pend),
                                               rsrc, (int)off, 0, WT ? /*sc0 sc1*/ 17 : 0);
      }
      if (flags & KT_FIRST)
      {
        acc0 = bv + mv * cond;
        acc1 = f4{0.f, 0.f, 0.f, 0.f};
      }
      // one wait for the whole set (the oldest requests in flight), not one per tap
#pragma unroll
      for (int i = 0; i < kKtTaps; i++)
        asm volatile("" ::"v"(s.th[i]));
#pragma unroll
      for (int i = 0; i < NT; i++)
        asm volatile("" ::"v"(s.tt[i]));
      auto tap = [&](int i) {
        const fN bsum = lv[i] + s.th[i]; // exactly one of the two is the operand, the other is 0
#pragma unroll
        for (int m = 0; m < NK; m++)
        {
          const float bm = bsum[m];
          const float am = NK == 2 ? s.tt[i / 2][(i % 2) * 2 + m] : s.tt[i][m];
          if (i & 1)
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(am, bm, acc1, 0, 0, 0);
          else
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(am, bm, acc0, 0, 0, 0);
        }
      };
      if (ntaps == kKtTaps) // the common case (A2: 26 of 30 chunks) without a test per tap
      {
#pragma unroll
        for (int i = 0; i < kKtTaps; i++)
          tap(i);
      }
      else
      {
#pragma unroll
        for (int i = 0; i < kKtTaps; i++)
          if (i < ntaps)
            tap(i);
      }
      const float cn = s.cn;
      // refill this set for chunk ci + D (beyond the end of this block it belongs to the next one, whose rings have
      // moved on by one block)
      fetch(s, Dq, ci + D >= NCH ? 1 : 0);
      if (flags & KT_LAST)
      {
        const f4 pre = acc0 + acc1;
        f4 pub;
        if (flags & KT_HEAD)
        {
          if (out && g == 0 && frame < nvalid)
            out[(size_t)blk * kBlock + frame] = head_scale * pre[0];
          // block boundary: ring write positions move on, the next block's layer 0 input is rechannel * sample
          wposv += nvalid;
          if (wposv >= ring_len_v)
            wposv -= ring_len_v;
          blk++;
          nvalid = min(kBlock, a.n_frames - blk * kBlock);
          cond = cn;
          x = rech * cond;
          head = f4{0.f, 0.f, 0.f, 0.f};
          pub = x;
        }
        else
        {
          const f4 z = mf::act4<ACT_T>(act, pre, act_p0);
          head += z;
          f4 y0 = x + b1v, y1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int m = 0; m < NK; m += 2)
          {
            y0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[m], z[m], y0, 0, 0, 0);
            y1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[m + 1], z[m + 1], y1, 0, 0, 0);
          }
          x = y0 + y1;
          pub = (flags & KT_NEXT_HEAD) ? head : x;
        }
        par ^= 1u;
        if (pub_lane)
          mf::lds_st4(lds, par * buf_b + (unsigned)(frame + 1) * row_b + quad_b, pub);
        pend = pub;
        mf::lds_barrier();
      }
      if (active && ++ci == NCH)
        ci = 0;
    }
  }
  if (w == 0 && lane < a.n_rings)
    wpos_tbl[lane] = wposv;
}

namespace
{
template <int ACT_T, bool WT, bool PROF, int D>
hipError_t launch_mfma_inst(const A1Args& a, int n_blocks, hipStream_t stream)
{
  static int lds_limit = 0; // per instantiation: dynamic LDS the runtime has been told about
  if (a.lds_bytes > lds_limit)
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&nam_a1_mfma_kernel<ACT_T, WT, PROF, D>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, a.lds_bytes);
    if (e != hipSuccess)
      return e;
    lds_limit = a.lds_bytes;
  }
  hipLaunchKernelGGL((nam_a1_mfma_kernel<ACT_T, WT, PROF, D>), dim3(n_blocks), dim3(512), a.lds_bytes, stream, a.plan,
                     a.blob, a);
  return hipGetLastError();
}
template <int ACT_T, bool WT, bool PROF>
hipError_t launch_mfma_depth(const A1Args& a, int n_blocks, hipStream_t stream)
{
  return a.prefetch == 5 ? launch_mfma_inst<ACT_T, WT, PROF, 5>(a, n_blocks, stream)
                         : launch_mfma_inst<ACT_T, WT, PROF, 6>(a, n_blocks, stream);
}
} // namespace

hipError_t launch_a1_mfma(const A1Args& a, int n_blocks, int act, hipStream_t stream)
{
  const bool wt = a.n_frames <= 2 * kBlock; // short launches write ring appends through (see ring_store)
  if (a.prefetch != 5 && a.prefetch != 6)
    return hipErrorInvalidValue;
  if (a.dbg) // developer tool: barrier-wait profile of workgroup 0
    return act == ACT_FASTTANH ? launch_mfma_depth<ACT_FASTTANH, false, true>(a, n_blocks, stream)
                               : launch_mfma_depth<-1, false, true>(a, n_blocks, stream);
  if (act == ACT_FASTTANH)
    return wt ? launch_mfma_depth<ACT_FASTTANH, true, false>(a, n_blocks, stream)
              : launch_mfma_depth<ACT_FASTTANH, false, false>(a, n_blocks, stream);
  if (act == ACT_TANH)
    return wt ? launch_mfma_depth<ACT_TANH, true, false>(a, n_blocks, stream)
              : launch_mfma_depth<ACT_TANH, false, false>(a, n_blocks, stream);
  return wt ? launch_mfma_depth<-1, true, false>(a, n_blocks, stream) : launch_mfma_depth<-1, false, false>(a, n_blocks, stream);
}

hipError_t launch_kt_mfma(const A1Args& a, int n_blocks, int nk, int channels, int lds_aux_floats, int act,
                          hipStream_t stream)
{
  const bool wt = a.n_frames <= 2 * kBlock; // short launches write ring appends through (see ring_store)
  const int lds_bytes = (2 * (kBlock + 1) * (channels + 4) + lds_aux_floats) * (int)sizeof(float);
  if (lds_bytes > 64 * 1024)
    return hipErrorInvalidValue; // (plan.cpp keeps the LDS copy small; 16-channel models with 32 layers stay below)
#define NAM_KT(NK, WT, ACT) \
  hipLaunchKernelGGL((nam_kt_mfma_kernel<NK, WT, ACT>), dim3(n_blocks), dim3(256), lds_bytes, stream, a.plan, a.blob, a)
#define NAM_KT_ACT(NK, WT) \
  switch (act) \
  { \
    case ACT_LEAKYRELU: NAM_KT(NK, WT, ACT_LEAKYRELU); break; \
    case ACT_FASTTANH: NAM_KT(NK, WT, ACT_FASTTANH); break; \
    case ACT_TANH: NAM_KT(NK, WT, ACT_TANH); break; \
    default: NAM_KT(NK, WT, -1); break; \
  }
  if (nk == 2)
  {
    if (wt)
      NAM_KT_ACT(2, true)
    else
      NAM_KT_ACT(2, false)
  }
  else
  {
    if (wt)
      NAM_KT_ACT(4, true)
    else
      NAM_KT_ACT(4, false)
  }
#undef NAM_KT_ACT
#undef NAM_KT
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

hipError_t launch_lstm_mfma(const LSTMArgs& a, hipStream_t stream)
{
  static int lds_limit = 0;
  if (a.mf_lds_bytes > lds_limit)
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(nam_lstm_mfma_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, a.mf_lds_bytes);
    if (e != hipSuccess)
      return e;
    lds_limit = a.mf_lds_bytes;
  }
  const int n_blocks = (a.n_streams + 15) / 16;
  // small models: everything in registers (only the I/O tiles in LDS)
  if (a.input_size <= 4 && a.n_layers <= 2 && a.mf_nt <= 6)
  {
    // I/O tiles + a pad row that lanes without an output row store to (64 lanes + 64 steps)
    const int io_bytes = ((a.in_ch + a.out_ch) * 16 * 65 + 128) * (int)sizeof(float);
#define NAM_LSTM_REG(NL, NT) \
  hipLaunchKernelGGL((nam_lstm_mfma_reg_kernel<NL, NT>), dim3(n_blocks), dim3(64), io_bytes, stream, a.blob, a)
    const int key = a.n_layers * 10 + a.mf_nt;
    switch (key)
    {
      case 11: NAM_LSTM_REG(1, 1); break;
      case 12: NAM_LSTM_REG(1, 2); break;
      case 13: NAM_LSTM_REG(1, 3); break;
      case 14: NAM_LSTM_REG(1, 4); break;
      case 21: NAM_LSTM_REG(2, 1); break;
      case 22: NAM_LSTM_REG(2, 2); break;
      case 23: NAM_LSTM_REG(2, 3); break;
      case 24: NAM_LSTM_REG(2, 4); break;
      case 15: NAM_LSTM_REG(1, 5); break;
      case 16: NAM_LSTM_REG(1, 6); break;
      case 25: NAM_LSTM_REG(2, 5); break;
      default: NAM_LSTM_REG(2, 6); break;
    }
#undef NAM_LSTM_REG
    return hipGetLastError();
  }
  hipLaunchKernelGGL(nam_lstm_mfma_kernel, dim3(n_blocks), dim3(64), a.mf_lds_bytes, stream, a.blob, a);
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
