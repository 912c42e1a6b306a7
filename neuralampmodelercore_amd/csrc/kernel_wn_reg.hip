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

#include <cstddef>
#include <type_traits>

#include "device_common.h"
#include "kernels.h"
#include "persist_wave.h"

namespace namhip
{
namespace
{
using mf::f4;
using mf::lds_ld4;
using mf::lds_st4;

// the fields of a WrOp the kernel reads, as scalars (a whole-struct copy would park the float and the padding in scratch)
struct WrOpS
{
  int type, shape, w, hist, dil, flags, act, act2, n_out, scale_bits;
  __device__ __forceinline__ float scale() const { return __builtin_bit_cast(float, scale_bits); }
};
__device__ __forceinline__ WrOpS wr_fetch(const WrOp* ops, int i)
{
  const int* p = reinterpret_cast<const int*>(ops + i);
  static_assert(offsetof(WrOp, scale) == 44 && offsetof(WrOp, n_out) == 40 && offsetof(WrOp, act) == 28, "WrOp layout");
  return WrOpS{p[0], p[1], p[2], p[3], p[5], p[6], p[7], p[8], p[10], p[11]};
}

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

using f2 = __attribute__((ext_vector_type(2))) float;

// Accumulators are PAIRS of outputs in adjacent registers (f2), so that one v_pk_fma_f32 advances two outputs; a
// vector of N values occupies pad2(N) / 2 pairs (the odd tail is padding: zero weights, zero bias, never read).
constexpr int wr_pairs(int n)
{
  return (n + 1) / 2;
}
__device__ __forceinline__ float pget(const f2* v, int i)
{
  return v[i >> 1][i & 1];
}
template <int N>
__device__ __forceinline__ void wr_unpack(float* dst, const f2* src)
{
#pragma unroll
  for (int i = 0; i < N; i++)
    dst[i] = pget(src, i);
}
template <int N>
__device__ __forceinline__ void wr_pack(f2* dst, const float* src)
{
#pragma unroll
  for (int i = 0; i < wr_pairs(N); i++)
    dst[i] = f2{src[2 * i], 2 * i + 1 < N ? src[2 * i + 1] : 0.0f};
}

// Weights travel LDS -> registers one step AHEAD of the arithmetic that uses them: a step's b128 reads are issued (and
// pinned in place by wr_fence) before the previous step's FMAs, so that their ~100+ cycle latency runs under those FMAs
// instead of in front of every pair of them (one wavefront per SIMD: nothing else hides it).
__device__ __forceinline__ void wr_fence()
{
  asm volatile("" ::: "memory"); // LDS reads issued so far stay above this point; register arithmetic is free to move
}

// a matrix stored transposed [IN][pad4(OUT)] (+ an optional bias [pad4(OUT)]): one b128 = four outputs' weights for
// one input = two packed FMAs with the input on both halves
template <int OUT, int IN>
struct WrMat
{
  f4 w[IN][wr_pad4(OUT) / 4];
  f4 b[wr_pad4(OUT) / 4];
};
template <int OUT, int IN>
__device__ __forceinline__ void wr_ld(WrMat<OUT, IN>& m, const char* lds, unsigned wb, bool bias, unsigned bb)
{
  constexpr int Q = wr_pad4(OUT) / 4;
#pragma unroll
  for (int i = 0; i < IN; i++)
#pragma unroll
    for (int q = 0; q < Q; q++)
      m.w[i][q] = lds_ld4(lds, wb + (unsigned)(i * Q + q) * 16u);
#pragma unroll
  for (int q = 0; q < Q; q++)
    m.b[q] = bias ? lds_ld4(lds, bb + (unsigned)q * 16u) : f4{0.f, 0.f, 0.f, 0.f};
}
// acc (pairs) = b + W in
template <int OUT, int IN>
__device__ __forceinline__ void wr_mv(f2* acc, const float* in, const WrMat<OUT, IN>& m)
{
  constexpr int Q = wr_pad4(OUT) / 4, P = wr_pairs(OUT);
#pragma unroll
  for (int q = 0; q < Q; q++)
  {
    acc[2 * q] = f2{m.b[q][0], m.b[q][1]};
    if (2 * q + 1 < P)
      acc[2 * q + 1] = f2{m.b[q][2], m.b[q][3]};
  }
#pragma unroll
  for (int i = 0; i < IN; i++)
  {
    const f2 xs = f2{in[i], in[i]};
#pragma unroll
    for (int q = 0; q < Q; q++)
    {
      acc[2 * q] = __builtin_elementwise_fma(f2{m.w[i][q][0], m.w[i][q][1]}, xs, acc[2 * q]);
      if (2 * q + 1 < P)
        acc[2 * q + 1] = __builtin_elementwise_fma(f2{m.w[i][q][2], m.w[i][q][3]}, xs, acc[2 * q + 1]);
    }
  }
}

// film.h:76-204 — v[d] = v[d] * scale[d] (+ shift[d]);  scale = Ws cond + bs, shift = Wh cond + bh
// block at `fb`: Ws [COND][pad4(D)], Wh [COND][pad4(D)], bs [pad4(D)], bh [pad4(D)]
template <int D, int COND>
struct WrFilm
{
  WrMat<(D > 0 ? D : 1), COND> s, h;
};
template <int D, int COND>
__device__ __forceinline__ void wr_ld(WrFilm<D, COND>& f, const char* lds, unsigned fb, bool shift)
{
  if constexpr (D > 0)
  {
    constexpr unsigned kMat = (unsigned)COND * wr_pad4(D) * 4u, kVec = (unsigned)wr_pad4(D) * 4u;
    wr_ld(f.s, lds, fb, true, fb + 2u * kMat);
    if (shift)
      wr_ld(f.h, lds, fb + kMat, true, fb + 2u * kMat + kVec);
  }
}
template <int D, int COND>
__device__ __forceinline__ void wr_film(f2* v, const float* cond, const WrFilm<D, COND>& f, bool shift)
{
  if constexpr (D > 0)
  {
    f2 sc[wr_pairs(D)];
    wr_mv(sc, cond, f.s);
    if (shift)
    {
      f2 sh[wr_pairs(D)];
      wr_mv(sh, cond, f.h);
#pragma unroll
      for (int d = 0; d < wr_pairs(D); d++)
        v[d] = __builtin_elementwise_fma(v[d], sc[d], sh[d]);
    }
    else
    {
#pragma unroll
      for (int d = 0; d < wr_pairs(D); d++)
        v[d] *= sc[d];
    }
  }
}

// activation parameters: p0..p3 and one PReLU slope per row
template <int N>
struct WrActP
{
  f4 p;
  f4 slope[wr_pad4(N) / 4];
};
template <int N>
__device__ __forceinline__ void wr_ld(WrActP<N>& a, const char* lds, unsigned ab)
{
  a.p = lds_ld4(lds, ab);
#pragma unroll
  for (int q = 0; q < wr_pad4(N) / 4; q++)
    a.slope[q] = lds_ld4(lds, ab + 16u + (unsigned)q * 16u);
}
// v[c] = act(v[c]) for rows [R0, R0 + N) of the pair vector: compile-time type (ACT >= 0) or one dispatch on the
// wavefront-uniform run-time type, then straight-line code
template <int N, int R0, int ACT>
__device__ __forceinline__ void wr_act(int type, f2* v, const WrActP<N>& ap)
{
  if constexpr (ACT == ACT_IDENTITY)
    return;
  if (ACT < 0 && type == ACT_IDENTITY)
    return;
  const f4 p = ap.p;
  auto run = [&](auto tag) {
    constexpr int T = decltype(tag)::value;
#pragma unroll
    for (int c = 0; c < N; c++)
      v[(R0 + c) >> 1][(R0 + c) & 1] = d_act<T>(pget(v, R0 + c), p[0], p[1], p[2], p[3], ap.slope[c >> 2][c & 3]);
  };
  if constexpr (ACT >= 0)
    run(std::integral_constant<int, ACT>{});
  else
  {
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
}

// _Layer::process, model.cpp:183-393 (the oracle's orc_layer_process walks the same steps; the mixin path, which does
// not depend on the conv, is moved in front of it so that it runs while the conv's taps come back from LDS).
// FM >= 0: FiLM mask / shift mask / blend / activation types are compile-time (one straight-line block); FM < 0: run-time
// flags from the op
template <int COND, int C, int B, bool G, int K, int HO, int FM, int SM, int BL, int A1, int A2>
__device__ __forceinline__ void wr_layer(WrRegs& r, const WrOpS& op, char* lds, int lane)
{
  constexpr WrLayerLayout L = wr_layer_layout(COND, C, B, G, K, HO);
  constexpr int ZC = G ? 2 * B : B;
  const unsigned wb = (unsigned)op.w * 4u;
  const int fl = FM >= 0 ? (FM | (SM << 8) | (BL << 16)) : op.flags;
  auto on = [&](int slot) { return ((fl >> slot) & 1) != 0; };
  auto sh = [&](int slot) { return ((fl >> (8 + slot)) & 1) != 0; };
  auto fb = [&](int slot) { return wb + (unsigned)L.film[slot] * 4u; };
  const bool blended = (fl & (1 << 16)) != 0;

  WrFilm<C, COND> f_cpre;
  WrFilm<COND, COND> f_mpre;
  WrMat<ZC, COND> m_mix;
  WrFilm<ZC, COND> f_mpost, f_cpost, f_apre;
  WrMat<ZC, K * C> m_conv;
  WrFilm<B, COND> f_apost;
  WrActP<G ? B : ZC> a_1;
  WrActP<B> a_2;
  WrMat<C, B> m_l1;
  WrFilm<C, COND> f_l1;
  WrMat<(HO > 0 ? HO : 1), B> m_h1;
  WrFilm<HO, COND> f_h1;

  if (on(FILM_CONV_PRE))
    wr_ld(f_cpre, lds, fb(FILM_CONV_PRE), sh(FILM_CONV_PRE));
  if (on(FILM_MIXIN_PRE))
    wr_ld(f_mpre, lds, fb(FILM_MIXIN_PRE), sh(FILM_MIXIN_PRE));
  wr_fence();
  wr_ld(m_mix, lds, wb + L.mixin * 4u, false, 0u);
  wr_fence();

  // Step 1a: the conv's input (pre FiLM) goes into its history rows; the taps are requested — model.cpp:189-203
  float ci[C];
  if (on(FILM_CONV_PRE))
  {
    f2 t[wr_pairs(C)];
    wr_pack<C>(t, r.x);
    wr_film<C, COND>(t, r.cond, f_cpre, sh(FILM_CONV_PRE));
    wr_unpack<C>(ci, t);
  }
  else
  {
#pragma unroll
    for (int i = 0; i < C; i++)
      ci[i] = r.x[i];
  }
  const unsigned hb = ((unsigned)op.hist + 64u + (unsigned)lane) * 4u; // this block's frames, behind the 64 before them
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
  if (on(FILM_MIXIN_POST))
    wr_ld(f_mpost, lds, fb(FILM_MIXIN_POST), sh(FILM_MIXIN_POST));
  wr_fence();

  // input mixin (+ pre / post FiLM) — model.cpp:205-219
  float mi[COND];
  if (on(FILM_MIXIN_PRE))
  {
    f2 t[wr_pairs(COND)];
    wr_pack<COND>(t, r.cond);
    wr_film<COND, COND>(t, r.cond, f_mpre, sh(FILM_MIXIN_PRE));
    wr_unpack<COND>(mi, t);
  }
  else
  {
#pragma unroll
    for (int i = 0; i < COND; i++)
      mi[i] = r.cond[i];
  }
  wr_ld(m_conv, lds, wb + L.conv * 4u, true, wb + L.conv_b * 4u);
  wr_fence();
  f2 m[wr_pairs(ZC)];
  wr_mv(m, mi, m_mix);
  if (on(FILM_CONV_POST))
    wr_ld(f_cpost, lds, fb(FILM_CONV_POST), sh(FILM_CONV_POST));
  wr_fence();
  if (on(FILM_MIXIN_POST))
    wr_film<ZC, COND>(m, r.cond, f_mpost, sh(FILM_MIXIN_POST));
  if (on(FILM_ACT_PRE))
    wr_ld(f_apre, lds, fb(FILM_ACT_PRE), sh(FILM_ACT_PRE));
  wr_fence();

  // Step 1b: the convolution (+ post FiLM); z = conv + mixin — model.cpp:189-221
  f2 z[wr_pairs(ZC)];
  wr_mv(z, taps, m_conv);
  wr_ld(a_1, lds, wb + L.act * 4u);
  if constexpr (G)
    wr_ld(a_2, lds, wb + L.act2 * 4u);
  if (on(FILM_ACT_POST))
    wr_ld(f_apost, lds, fb(FILM_ACT_POST), sh(FILM_ACT_POST));
  wr_fence();
  if (on(FILM_CONV_POST))
    wr_film<ZC, COND>(z, r.cond, f_cpost, sh(FILM_CONV_POST));
#pragma unroll
  for (int c = 0; c < wr_pairs(ZC); c++)
    z[c] += m[c];
  wr_ld(m_l1, lds, wb + L.l1 * 4u, true, wb + L.l1_b * 4u);
  wr_fence();
  if (on(FILM_ACT_PRE))
    wr_film<ZC, COND>(z, r.cond, f_apre, sh(FILM_ACT_PRE));

  // Steps 2 and 3: activation (+ gating / blending) and the 1x1 — model.cpp:234-288
  if constexpr (!G)
    wr_act<ZC, 0, A1>(op.act, z, a_1);
  else
  {
    // gating_activations.h:59-114 (gated: a * g) / :165-228 (blended: alpha * a + (1 - alpha) * pre)
    float pre[B];
    wr_unpack<B>(pre, z);
    wr_act<B, 0, A1>(op.act, z, a_1);
    wr_act<B, B, A2>(op.act2, z, a_2);
#pragma unroll
    for (int c = 0; c < B; c++)
    {
      const float a = pget(z, c), g = pget(z, B + c);
      z[c >> 1][c & 1] = blended ? __builtin_fmaf(g, a, (1.0f - g) * pre[c]) : a * g;
    }
  }
  if constexpr (HO > 0)
    wr_ld(m_h1, lds, wb + L.h1 * 4u, true, wb + L.h1_b * 4u);
  wr_fence();
  if (on(FILM_ACT_POST))
  {
    if constexpr (G && (B & 1)) // (an odd B would share its last pair with a gate row: not instantiated)
      __builtin_trap();
    wr_film<B, COND>(z, r.cond, f_apost, sh(FILM_ACT_POST));
  }
  float zb[B]; // the B rows the 1x1s read
  wr_unpack<B>(zb, z);
  const bool l1_film = G && blended && on(FILM_LAYER1X1_POST); // quirk kept: only in the BLENDED branch, model.cpp:282-286
  if (l1_film)
    wr_ld(f_l1, lds, fb(FILM_LAYER1X1_POST), sh(FILM_LAYER1X1_POST));
  if (HO > 0 && on(FILM_HEAD1X1_POST))
    wr_ld(f_h1, lds, fb(FILM_HEAD1X1_POST), sh(FILM_HEAD1X1_POST));
  wr_fence();
  f2 l1[wr_pairs(C)];
  wr_mv(l1, zb, m_l1);
  if (l1_film)
    wr_film<C, COND>(l1, r.cond, f_l1, sh(FILM_LAYER1X1_POST));

  // head contribution — model.cpp:290-352, accumulated by the array (:513-531)
  if constexpr (HO > 0)
  {
    f2 h[wr_pairs(HO)];
    wr_mv(h, zb, m_h1);
    if (on(FILM_HEAD1X1_POST))
      wr_film<HO, COND>(h, r.cond, f_h1, sh(FILM_HEAD1X1_POST));
#pragma unroll
    for (int c = 0; c < HO; c++)
      r.hacc[c] += pget(h, c);
  }
  else
  {
#pragma unroll
    for (int c = 0; c < B; c++)
      r.hacc[c] += zb[c];
  }
  // residual — model.cpp:354-392
#pragma unroll
  for (int i = 0; i < C; i++)
    r.x[i] += pget(l1, i);
}

// _LayerArray::process prologue, model.cpp:463-492: the head accumulator starts from the previous array's head output
// (or zero), the rechannel 1x1 (no bias) maps the previous array's layer output (or the model input) to C channels
template <int IN, int C>
__device__ __forceinline__ void wr_array_begin(WrRegs& r, const WrOpS& op, const char* lds)
{
  const bool first = (op.flags & 1) != 0;
#pragma unroll
  for (int c = 0; c < kWrRegs; c++)
    r.hacc[c] = first ? 0.0f : r.hout[c];
  float src[IN];
#pragma unroll
  for (int i = 0; i < IN; i++)
    src[i] = first ? r.in[i] : r.x[i];
  WrMat<C, IN> m; // [IN][pad4(C)], no bias
  wr_ld(m, lds, (unsigned)op.w * 4u, false, 0u);
  f2 xn[wr_pairs(C)];
  wr_mv(xn, src, m);
  wr_unpack<C>(r.x, xn);
}

// head rechannel (kernel size 1), model.cpp:547-548: head output = W head accumulator (+ bias); W^T = [HI][pad4(HS)]
template <int HI, int HS>
__device__ __forceinline__ void wr_array_end(WrRegs& r, const WrOpS& op, const char* lds)
{
  WrMat<HS, HI> m;
  wr_ld(m, lds, (unsigned)op.w * 4u, true, (unsigned)op.w * 4u + (unsigned)(HI * wr_pad4(HS)) * 4u); // (a zero bias when none)
  f2 o[wr_pairs(HS)];
  wr_mv(o, r.hacc, m);
  wr_unpack<HS>(r.hout, o);
}

} // namespace

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void nam_wn_reg_kernel(const WrArgs a)
{
  extern __shared__ __attribute__((aligned(16))) char lds_wr[];
  char* const lds = lds_wr;
  const int lane = (int)threadIdx.x;
  const int stream = a.stream_map ? a.stream_map[blockIdx.x] : (int)blockIdx.x;
  // persistent session (persist_wave.h): blocks come from commands, not from a frame count
  const bool pers = a.ps.ring != nullptr;
  PersistWave pw;
  unsigned cmd_off = 0;
  if (pers && !pw.begin(a.ps, (int)blockIdx.x, cmd_off))
  {
    pw.leave(a.ps, (int)blockIdx.x); // nothing to do
    return;
  }
  // weights -> LDS (the blob is a multiple of 4 floats) and conv input histories <- state (row r = the last 64 frames
  // of one channel of one layer's conv input): eight requests in flight per round trip to memory
  float* const st = a.state + (long)stream * a.state_stride;
  const unsigned hist0 = (unsigned)a.hist_base * 4u;
  for (int base = 0; base < a.blob_floats; base += 8 * 256)
  {
    f4 v[8];
#pragma unroll
    for (int u = 0; u < 8; u++)
    {
      const int i = base + u * 256 + lane * 4;
      v[u] = i < a.blob_floats ? *reinterpret_cast<const f4*>(a.blob + i) : f4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 8; u++)
    {
      const int i = base + u * 256 + lane * 4;
      if (i < a.blob_floats)
        lds_st4(lds, (unsigned)i * 4u, v[u]);
    }
  }
  for (int base = 0; base < a.n_rows; base += 8)
  {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; u++)
      v[u] = base + u < a.n_rows ? st[(base + u) * 64 + lane] : 0.0f;
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (base + u < a.n_rows)
        lds_st1(lds, hist0 + (unsigned)((base + u) * kWrPitch + lane) * 4u, v[u]);
  }
  const float* const in = a.in ? a.in + (long)stream * a.in_ch * a.io_stride : nullptr;
  float* const out = a.out ? a.out + (long)stream * a.out_ch * a.io_stride : nullptr;

  for (int f0 = pers ? (int)cmd_off : 0;;)
  {
    const int n = pers ? kBlock : min(kBlock, a.n_frames - f0);
    if (pers)
      pw.look_ahead(a.ps);
    WrRegs r;
#pragma unroll
    for (int c = 0; c < kWrRegs; c++)
    {
      r.in[c] = 0.0f;
      if (in && c < a.in_ch && lane < n)
        r.in[c] = pers ? persist_in(in + (long)c * a.io_stride + f0 + lane) : in[(long)c * a.io_stride + f0 + lane];
      r.cond[c] = r.in[c]; // a net without condition_dsp (and the nested net itself) is conditioned on its input
      r.x[c] = r.hacc[c] = r.hout[c] = 0.0f;
    }
    WrOpS cur = wr_fetch(a.ops, 0);
    for (int oi = 0; oi < a.n_ops; oi++)
    {
      const WrOpS nxt = wr_fetch(a.ops, min(oi + 1, a.n_ops - 1)); // requested before this op runs
      switch (cur.type)
      {
        case WR_LAYER:
          switch (cur.shape)
          {
#define X(ID, COND, C, B, G, K, HO, FM, SM, BL, A1, A2) \
  case ID: wr_layer<COND, C, B, G, K, HO, FM, SM, BL, A1, A2>(r, cur, lds, lane); break;
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
            r.cond[c] = cur.scale() * r.hout[c];
          break;
        case WR_OUTPUT:
          if (out)
          {
#pragma unroll
            for (int c = 0; c < kWrRegs; c++)
              if (c < cur.n_out && lane < n)
                out[(long)c * a.io_stride + f0 + lane] = cur.scale() * r.hout[c];
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
    if (pers)
    {
      if (!pw.next(a.ps, (int)blockIdx.x, cmd_off))
        break; // ring empty: leave
      f0 = (int)cmd_off;
    }
    else
    {
      f0 += kBlock;
      if (f0 >= a.n_frames)
        break;
    }
  }
  for (int row = 0; row < a.n_rows; row++)
    st[row * 64 + lane] = lds_ld1(lds, hist0 + (unsigned)(row * kWrPitch + lane) * 4u);
  if (pers)
    pw.leave(a.ps, (int)blockIdx.x);
}

hipError_t launch_wn_reg(const WrArgs& a, int n_streams, int lds_bytes, hipStream_t stream)
{
  if (n_streams <= 0 || a.n_frames <= 0)
    return hipSuccess;
  hipLaunchKernelGGL(nam_wn_reg_kernel, dim3(n_streams), dim3(64), lds_bytes, stream, a);
  return hipGetLastError();
}

} // namespace namhip
