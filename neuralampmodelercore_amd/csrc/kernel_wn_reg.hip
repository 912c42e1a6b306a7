// kernel_wn_reg.hip — nam_wn_reg_kernel: WaveNets of a few channels per layer with every activation in registers.
//
// What it replaces in the reference: nam::wavenet::WaveNet::process (NAM/wavenet/model.cpp:822-910) for models like
// example_models/wavenet_a2_max.nam — `_process_condition` with a nested condition_dsp (:777-807), `_LayerArray::process`
// (:463-549) and `_Layer::process` (:183-393) with all eight FiLM slots (NAM/film.h:76-204), gated / blended
// activations (NAM/gating_activations.h:59-228), grouped 1x1s and head1x1 — 56 FiLMs and ~1,600 MACs per sample spread
// over ~300 matrix operations of 1..8 rows, which the op interpreter (kernel_generic.hip) spends 60 k instructions per
// 64-frame block on, nearly all of it dispatch and LDS row traffic.
//
// Mapping (plan.h: WrPlan): one wavefront per stream, lane = frame of the 64-frame block. A WaveNet has no recurrence —
// every frame of a block is independent given the conv inputs of the previous 64 frames — so a lane carries its frame
// through the whole network: the layer input x[C], the condition, the head accumulator and head output live in
// registers from the input sample to the output sample. A layer is ONE fully unrolled function, instantiated per
// (condition size, channels, bottleneck, gating, kernel size, head1x1 size) shape; weights are read from an LDS copy of
// the blob as broadcast b128 reads (every lane the same address: no bank conflicts, 4 weights per instruction). The only
// per-frame LDS traffic is the conv input: each layer stores its C values and reads (K - 1) * C taps from the rows its
// neighbours (and the previous block, kept 64 frames back) wrote. One wavefront per workgroup: LDS operations of a
// wavefront execute in order, so no barrier anywhere.
//
// FiLM slots and the shift are run-time flags (wavefront-uniform branches); activation types are run-time (one
// dispatch per layer and activation, not per channel); grouped convs arrive expanded to dense from the planner.

#include <type_traits>

#include "device_common.h"
#include "kernels.h"

namespace namhip
{
namespace
{
using mf::f4;
using mf::lds_ld4;
using mf::lds_st4;

struct WrRegs
{
  float in[kWrRegs]; // the model's input sample (every channel)
  float x[kWrRegs]; // layer input / output
  float cond[kWrRegs]; // condition signal
  float hacc[kWrRegs]; // head accumulator of the current array
  float hout[kWrRegs]; // head rechannel output of the last finished array
};

__device__ __forceinline__ float lds_ld1(const char* lds, unsigned byte_off)
{
  return *reinterpret_cast<const float*>(lds + byte_off);
}
__device__ __forceinline__ void lds_st1(char* lds, unsigned byte_off, float v)
{
  *reinterpret_cast<float*>(lds + byte_off) = v;
}

// dst[0..N) = the N floats at byte offset `off` (rows are padded to 4 floats: whole b128 reads)
template <int N>
__device__ __forceinline__ void wr_load(float* dst, const char* lds, unsigned off)
{
#pragma unroll
  for (int i = 0; i < N; i += 4)
  {
    const f4 v = lds_ld4(lds, off + (unsigned)i * 4u);
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (i + j < N)
        dst[i + j] = v[j];
  }
}

// acc[o] += sum_i W[o][i] * in[i];  W = [OUT][pad4(IN)] at byte offset `wb`
template <int OUT, int IN>
__device__ __forceinline__ void wr_mv(float* acc, const float* in, const char* lds, unsigned wb)
{
  constexpr int IN4 = wr_pad4(IN);
#pragma unroll
  for (int o = 0; o < OUT; o++)
  {
#pragma unroll
    for (int i = 0; i < IN4; i += 4)
    {
      const f4 w = lds_ld4(lds, wb + (unsigned)(o * IN4 + i) * 4u);
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (i + j < IN)
          acc[o] = __builtin_fmaf(w[j], in[i + j], acc[o]);
    }
  }
}

// film.h:76-204 — v[d] = v[d] * scale[d] (+ shift[d]);  [scale; shift] = W[2D][COND] cond + b
template <int D, int COND>
__device__ __forceinline__ void wr_film(float* v, const float* cond, const char* lds, unsigned fb, bool shift)
{
  if constexpr (D > 0)
  {
    constexpr unsigned kBias = 2u * D * wr_pad4(COND) * 4u;
    float ss[2 * D];
    wr_load<2 * D>(ss, lds, fb + kBias);
    wr_mv<D, COND>(ss, cond, lds, fb);
    if (shift)
    {
      wr_mv<D, COND>(ss + D, cond, lds, fb + (unsigned)D * wr_pad4(COND) * 4u);
#pragma unroll
      for (int d = 0; d < D; d++)
        v[d] = __builtin_fmaf(v[d], ss[d], ss[D + d]);
    }
    else
    {
#pragma unroll
      for (int d = 0; d < D; d++)
        v[d] *= ss[d];
    }
  }
}

// v[c] = act(v[c]) for c < N: one dispatch on the (wavefront-uniform) type, then straight-line code
template <int N>
__device__ __forceinline__ void wr_act(int type, float* v, const char* lds, unsigned ab)
{
  if (type == ACT_IDENTITY)
    return;
  const f4 p = lds_ld4(lds, ab);
  float slope[N];
  wr_load<N>(slope, lds, ab + 16u);
  auto run = [&](auto tag) {
    constexpr int T = decltype(tag)::value;
#pragma unroll
    for (int c = 0; c < N; c++)
      v[c] = d_act<T>(v[c], p[0], p[1], p[2], p[3], slope[c]);
  };
#define NAM_WR_ACT(T) \
  case T: run(std::integral_constant<int, T>{}); break;
  switch (type)
  {
    NAM_WR_ACT(ACT_TANH)
    NAM_WR_ACT(ACT_HARDTANH)
    NAM_WR_ACT(ACT_FASTTANH)
    NAM_WR_ACT(ACT_RELU)
    NAM_WR_ACT(ACT_LEAKYRELU)
    NAM_WR_ACT(ACT_PRELU)
    NAM_WR_ACT(ACT_SIGMOID)
    NAM_WR_ACT(ACT_SILU)
    NAM_WR_ACT(ACT_HARDSWISH)
    NAM_WR_ACT(ACT_LEAKYHARDTANH)
    NAM_WR_ACT(ACT_SOFTSIGN)
    NAM_WR_ACT(ACT_FASTSIGMOID)
    default: __builtin_trap(); // (LUT activations are not planned onto this kernel)
  }
#undef NAM_WR_ACT
}

// _Layer::process, model.cpp:183-393 (the oracle's orc_layer_process walks the same steps)
template <int COND, int C, int B, bool G, int K, int HO>
__device__ __forceinline__ void wr_layer(WrRegs& r, const WrOp& op, char* lds, int lane)
{
  constexpr WrLayerLayout L = wr_layer_layout(COND, C, B, G, K, HO);
  constexpr int ZC = G ? 2 * B : B;
  const unsigned wb = (unsigned)op.w * 4u;
  const int fl = op.flags;
  auto on = [&](int slot) { return (fl >> slot) & 1; };
  auto sh = [&](int slot) { return ((fl >> (8 + slot)) & 1) != 0; };

  // Step 1: input convolution (+ pre / post FiLM) — model.cpp:189-203
  float ci[C];
#pragma unroll
  for (int i = 0; i < C; i++)
    ci[i] = r.x[i];
  if (on(FILM_CONV_PRE))
    wr_film<C, COND>(ci, r.cond, lds, wb + L.film[FILM_CONV_PRE] * 4u, sh(FILM_CONV_PRE));
  // the conv's input history: this block's frames behind the 64 frames before them
  const unsigned hb = ((unsigned)op.hist + 64u + (unsigned)lane) * 4u;
#pragma unroll
  for (int i = 0; i < C; i++)
    lds_st1(lds, hb + (unsigned)i * (kWrPitch * 4u), ci[i]);
  float taps[K * C]; // [k][i]: tap k looks (K - 1 - k) * dilation frames back (conv1d.cpp: the last tap is "now")
#pragma unroll
  for (int k = 0; k + 1 < K; k++)
  {
    const unsigned a = hb - (unsigned)((K - 1 - k) * op.dil) * 4u;
#pragma unroll
    for (int i = 0; i < C; i++)
      taps[k * C + i] = lds_ld1(lds, a + (unsigned)i * (kWrPitch * 4u));
  }
#pragma unroll
  for (int i = 0; i < C; i++)
    taps[(K - 1) * C + i] = ci[i];
  float z[ZC];
  wr_load<ZC>(z, lds, wb + L.conv_b * 4u);
  wr_mv<ZC, K * C>(z, taps, lds, wb + L.conv * 4u);
  if (on(FILM_CONV_POST))
    wr_film<ZC, COND>(z, r.cond, lds, wb + L.film[FILM_CONV_POST] * 4u, sh(FILM_CONV_POST));

  // input mixin (+ pre / post FiLM) — model.cpp:205-219; z = conv + mixin — :220
  {
    float mi[COND];
#pragma unroll
    for (int i = 0; i < COND; i++)
      mi[i] = r.cond[i];
    if (on(FILM_MIXIN_PRE))
      wr_film<COND, COND>(mi, r.cond, lds, wb + L.film[FILM_MIXIN_PRE] * 4u, sh(FILM_MIXIN_PRE));
    float m[ZC];
#pragma unroll
    for (int c = 0; c < ZC; c++)
      m[c] = 0.0f;
    wr_mv<ZC, COND>(m, mi, lds, wb + L.mixin * 4u);
    if (on(FILM_MIXIN_POST))
      wr_film<ZC, COND>(m, r.cond, lds, wb + L.film[FILM_MIXIN_POST] * 4u, sh(FILM_MIXIN_POST));
#pragma unroll
    for (int c = 0; c < ZC; c++)
      z[c] += m[c];
  }
  if (on(FILM_ACT_PRE))
    wr_film<ZC, COND>(z, r.cond, lds, wb + L.film[FILM_ACT_PRE] * 4u, sh(FILM_ACT_PRE));

  // Steps 2 and 3: activation (+ gating / blending) and the 1x1 — model.cpp:234-288
  if constexpr (!G)
    wr_act<ZC>(op.act, z, lds, wb + L.act * 4u);
  else
  {
    // gating_activations.h:59-114 (gated: a * g) / :165-228 (blended: alpha * a + (1 - alpha) * pre)
    float pre[B];
#pragma unroll
    for (int c = 0; c < B; c++)
      pre[c] = z[c];
    wr_act<B>(op.act, z, lds, wb + L.act * 4u);
    wr_act<B>(op.act2, z + B, lds, wb + L.act2 * 4u);
    const bool blended = (fl & (1 << 16)) != 0;
#pragma unroll
    for (int c = 0; c < B; c++)
      z[c] = blended ? __builtin_fmaf(z[B + c], z[c], (1.0f - z[B + c]) * pre[c]) : z[c] * z[B + c];
  }
  if (on(FILM_ACT_POST))
    wr_film<B, COND>(z, r.cond, lds, wb + L.film[FILM_ACT_POST] * 4u, sh(FILM_ACT_POST));
  float l1[C];
  wr_load<C>(l1, lds, wb + L.l1_b * 4u);
  wr_mv<C, B>(l1, z, lds, wb + L.l1 * 4u);
  if constexpr (G)
  {
    // quirk kept: layer1x1_post_film only runs in the BLENDED branch — model.cpp:282-286
    if ((fl & (1 << 16)) != 0 && on(FILM_LAYER1X1_POST))
      wr_film<C, COND>(l1, r.cond, lds, wb + L.film[FILM_LAYER1X1_POST] * 4u, sh(FILM_LAYER1X1_POST));
  }

  // head contribution — model.cpp:290-352, accumulated by the array (:513-531)
  if constexpr (HO > 0)
  {
    float h[HO];
    wr_load<HO>(h, lds, wb + L.h1_b * 4u);
    wr_mv<HO, B>(h, z, lds, wb + L.h1 * 4u);
    if (on(FILM_HEAD1X1_POST))
      wr_film<HO, COND>(h, r.cond, lds, wb + L.film[FILM_HEAD1X1_POST] * 4u, sh(FILM_HEAD1X1_POST));
#pragma unroll
    for (int c = 0; c < HO; c++)
      r.hacc[c] += h[c];
  }
  else
  {
#pragma unroll
    for (int c = 0; c < B; c++)
      r.hacc[c] += z[c];
  }
  // residual — model.cpp:354-392
#pragma unroll
  for (int i = 0; i < C; i++)
    r.x[i] += l1[i];
}

// _LayerArray::process prologue, model.cpp:463-492: the head accumulator starts from the previous array's head output
// (or zero), the rechannel 1x1 (no bias) maps the previous array's layer output (or the model input) to C channels
template <int IN, int C>
__device__ __forceinline__ void wr_array_begin(WrRegs& r, const WrOp& op, const char* lds)
{
  const bool first = (op.flags & 1) != 0;
#pragma unroll
  for (int c = 0; c < kWrRegs; c++)
    r.hacc[c] = first ? 0.0f : r.hout[c];
  float src[IN], xn[C];
#pragma unroll
  for (int i = 0; i < IN; i++)
    src[i] = first ? r.in[i] : r.x[i];
#pragma unroll
  for (int c = 0; c < C; c++)
    xn[c] = 0.0f;
  wr_mv<C, IN>(xn, src, lds, (unsigned)op.w * 4u);
#pragma unroll
  for (int c = 0; c < C; c++)
    r.x[c] = xn[c];
}

// head rechannel (kernel size 1), model.cpp:547-548: head output = W[HS][HI] head accumulator (+ bias)
template <int HI, int HS>
__device__ __forceinline__ void wr_array_end(WrRegs& r, const WrOp& op, const char* lds)
{
  float o[HS];
  wr_load<HS>(o, lds, (unsigned)op.w * 4u + (unsigned)(HS * wr_pad4(HI)) * 4u); // (zeros when there is no bias)
  wr_mv<HS, HI>(o, r.hacc, lds, (unsigned)op.w * 4u);
#pragma unroll
  for (int c = 0; c < HS; c++)
    r.hout[c] = o[c];
}

} // namespace

__global__ __launch_bounds__(64) void nam_wn_reg_kernel(const WrArgs a)
{
  extern __shared__ __attribute__((aligned(16))) char lds_wr[];
  char* const lds = lds_wr;
  const int lane = (int)threadIdx.x;
  const int stream = a.stream_map ? a.stream_map[blockIdx.x] : (int)blockIdx.x;
  // weights -> LDS (the blob is a multiple of 4 floats)
  for (int i = lane * 4; i < a.blob_floats; i += 256)
    lds_st4(lds, (unsigned)i * 4u, *reinterpret_cast<const f4*>(a.blob + i));
  // conv input histories <- state: row r = the last 64 frames of one channel of one layer's conv input
  float* const st = a.state + (long)stream * a.state_stride;
  const unsigned hist0 = (unsigned)a.hist_base * 4u;
  for (int row = 0; row < a.n_rows; row++)
    lds_st1(lds, hist0 + (unsigned)(row * kWrPitch + lane) * 4u, st[row * 64 + lane]);
  const float* const in = a.in ? a.in + (long)stream * a.in_ch * a.io_stride : nullptr;
  float* const out = a.out ? a.out + (long)stream * a.out_ch * a.io_stride : nullptr;

  for (int f0 = 0; f0 < a.n_frames; f0 += kBlock)
  {
    const int n = min(kBlock, a.n_frames - f0);
    WrRegs r;
#pragma unroll
    for (int c = 0; c < kWrRegs; c++)
    {
      r.in[c] = (in && c < a.in_ch && lane < n) ? in[(long)c * a.io_stride + f0 + lane] : 0.0f;
      r.cond[c] = r.in[c]; // a net without condition_dsp (and the nested net itself) is conditioned on its input
      r.x[c] = r.hacc[c] = r.hout[c] = 0.0f;
    }
    WrOp cur = a.ops[0];
    for (int oi = 0; oi < a.n_ops; oi++)
    {
      const WrOp nxt = a.ops[min(oi + 1, a.n_ops - 1)]; // requested before this op runs
      switch (cur.type)
      {
        case WR_LAYER:
          switch (cur.shape)
          {
#define X(ID, COND, C, B, G, K, HO) \
  case ID: wr_layer<COND, C, B, G, K, HO>(r, cur, lds, lane); break;
            WR_LAYER_SHAPES(X)
#undef X
            default: __builtin_trap();
          }
          break;
        case WR_ARRAY_BEGIN:
          switch (cur.shape)
          {
#define X(ID, IN, OUT) \
  case ID: wr_array_begin<IN, OUT>(r, cur, lds); break;
            WR_PAIR_SHAPES(X)
#undef X
            default: __builtin_trap();
          }
          break;
        case WR_ARRAY_END:
          switch (cur.shape)
          {
#define X(ID, IN, OUT) \
  case ID: wr_array_end<IN, OUT>(r, cur, lds); break;
            WR_PAIR_SHAPES(X)
#undef X
            default: __builtin_trap();
          }
          break;
        case WR_SET_COND:
#pragma unroll
          for (int c = 0; c < kWrRegs; c++)
            r.cond[c] = cur.scale * r.hout[c];
          break;
        case WR_OUTPUT:
          if (out)
          {
#pragma unroll
            for (int c = 0; c < kWrRegs; c++)
              if (c < cur.n_out && lane < n)
                out[(long)c * a.io_stride + f0 + lane] = cur.scale * r.hout[c];
          }
          break;
        default: __builtin_trap();
      }
      cur = nxt;
    }
    // the block's n frames move into the history: row[j] <- row[j + n] (reads of a wavefront precede its later writes)
    for (int row = 0; row < a.n_rows; row++)
    {
      const unsigned rb = hist0 + (unsigned)(row * kWrPitch) * 4u;
      const float v = lds_ld1(lds, rb + (unsigned)(lane + n) * 4u);
      lds_st1(lds, rb + (unsigned)lane * 4u, v);
    }
  }
  for (int row = 0; row < a.n_rows; row++)
    st[row * 64 + lane] = lds_ld1(lds, hist0 + (unsigned)(row * kWrPitch + lane) * 4u);
}

hipError_t launch_wn_reg(const WrArgs& a, int n_streams, int lds_bytes, hipStream_t stream)
{
  if (n_streams <= 0 || a.n_frames <= 0)
    return hipSuccess;
  hipLaunchKernelGGL(nam_wn_reg_kernel, dim3(n_streams), dim3(64), lds_bytes, stream, a);
  return hipGetLastError();
}

} // namespace namhip
