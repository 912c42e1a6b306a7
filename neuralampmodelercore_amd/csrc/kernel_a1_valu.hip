// kernel_a1_valu.hip — nam_a1_kernel: register-resident VALU kernel for the plain A1 family (see kernel_generic.hip for
// the file-level notes and the reference lines restated).
#include "device_common.h"

namespace namhip
{

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
  // number of output channels. K > 1 (A2: 16 taps): a single output channel (plan_a1.cpp); the accumulator goes
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

hipError_t launch_a1(const A1Args& a, int n_blocks, hipStream_t stream)
{
  hipLaunchKernelGGL(nam_a1_kernel, dim3(n_blocks), dim3(64), 0, stream, a.plan, a.blob, a);
  return hipGetLastError();
}

} // namespace namhip
