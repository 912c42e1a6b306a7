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
// every frame of a block is independent given the conv inputs of the earlier frames — so a lane carries its frame
// through the whole network: the layer input x[C], the condition, the head accumulator and head output live in
// registers from the input sample to the output sample. A layer is ONE fully unrolled function, instantiated per
// (condition size, channels, bottleneck, gating, kernel size, head1x1 size) shape; weights are read from an LDS copy of
// the blob as broadcast b128 reads (every lane the same address: no bank conflicts, 4 weights per instruction). The only
// per-frame LDS traffic is the conv input: each layer appends its C values to its LDS-resident ring (lookback + 64
// frames: the reference's RingBuffer, NAM/ring_buffer.cpp:7-109, kept on the CU) and reads (K - 1) * C taps from it. One
// wavefront per workgroup: LDS operations of a wavefront execute in order, so no barrier anywhere.
//
// A launch that runs more than one block (offline render, prewarm, a persistent session) brings the stream's whole ring
// area from its state in HBM and takes it back at the end; a launch of ONE block fetches only the 64-frame windows its
// taps reach (plan.h: the `pf` table) and stores only the frames it appended. One launch serves several width groups
// (kernels.h: WrArgs): a slimmable batch with mixed widths is ONE launch, and one persistent session.
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

using i4 = __attribute__((ext_vector_type(4))) int;

// the fields of a WrOp the kernel reads, as scalars (a whole-struct copy would park the float and the padding in scratch)
struct WrOpS
{
  int type, shape, w, hist, ring, dil, flags, act, act2, n_in, n_out, scale_bits, slot;
  __device__ __forceinline__ float scale() const { return __builtin_bit_cast(float, scale_bits); }
};
// op i of the program, from the LDS copy of the blob (every lane reads the same words: broadcast reads; the values go
// to scalar registers). From global memory an op cost a cache round trip (~0.5 us) that a small layer does not cover.
__device__ __forceinline__ WrOpS wr_fetch(const char* lds, unsigned ops_b, int i)
{
  static_assert(offsetof(WrOp, scale) == 44 && offsetof(WrOp, n_out) == 40 && offsetof(WrOp, act) == 28
                  && offsetof(WrOp, ring) == 16 && offsetof(WrOp, slot) == 48 && sizeof(WrOp) == 64,
                "WrOp layout");
  const unsigned a = ops_b + (unsigned)i * 64u;
  const i4 q0 = *reinterpret_cast<const i4*>(lds + a), q1 = *reinterpret_cast<const i4*>(lds + a + 16u),
           q2 = *reinterpret_cast<const i4*>(lds + a + 32u);
  const int q12 = *reinterpret_cast<const int*>(lds + a + 48u);
  auto u = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
  return WrOpS{u(q0[0]), u(q0[1]), u(q0[2]), u(q0[3]), u(q1[0]), u(q1[1]), u(q1[2]), u(q1[3]), u(q2[0]), u(q2[1]), u(q2[2]), u(q2[3]), u(q12)};
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


// A layer's conv-input ring (plan.h): [ceil(C / 4)][R][gs] floats, gs = 4 (the last group: C % 4) — a lane's frame of a
// group is one b128 / b64 / b32 access (three b32 for gs = 3), and the lane stride gs is bank-conflict free for each.
// `base_b`: byte offset of the area, `idx`: ring index of the lane's frame.
template <int C>
__device__ __forceinline__ void wr_ring_put(char* lds, unsigned base_b, int idx, int R, const float* v)
{
#pragma unroll
  for (int q = 0; q * 4 < C; q++)
  {
    const int gs = C - 4 * q < 4 ? C - 4 * q : 4;
    const unsigned a = base_b + (unsigned)(q * 4 * R + idx * gs) * 4u;
    if (gs == 4)
      lds_st4(lds, a, f4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]});
    else if (gs == 2)
      *reinterpret_cast<f2*>(lds + a) = f2{v[4 * q], v[4 * q + 1]};
    else
    {
#pragma unroll
      for (int i = 0; i < gs; i++)
        lds_st1(lds, a + (unsigned)i * 4u, v[4 * q + i]);
    }
  }
}
template <int C>
__device__ __forceinline__ void wr_ring_get(const char* lds, unsigned base_b, int idx, int R, float* v)
{
#pragma unroll
  for (int q = 0; q * 4 < C; q++)
  {
    const int gs = C - 4 * q < 4 ? C - 4 * q : 4;
    const unsigned a = base_b + (unsigned)(q * 4 * R + idx * gs) * 4u;
    if (gs == 4)
    {
      const f4 t = lds_ld4(lds, a);
      v[4 * q] = t[0], v[4 * q + 1] = t[1], v[4 * q + 2] = t[2], v[4 * q + 3] = t[3];
    }
    else if (gs == 2)
    {
      const f2 t = *reinterpret_cast<const f2*>(lds + a);
      v[4 * q] = t[0], v[4 * q + 1] = t[1];
    }
    else
    {
#pragma unroll
      for (int i = 0; i < gs; i++)
        v[4 * q + i] = lds_ld1(lds, a + (unsigned)i * 4u);
    }
  }
}

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

// acc (pairs) += W in  (no bias: the caller seeded acc)
template <int OUT, int IN>
__device__ __forceinline__ void wr_mv_acc(f2* acc, const float* in, const WrMat<OUT, IN>& m)
{
  constexpr int Q = wr_pad4(OUT) / 4, P = wr_pairs(OUT);
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

// A matrix in the MATRIX form (round 6): [lane % 4][pad4(OUT) / 4][pad4(IN)] — the lane's own row W[4 q + lane % 4][c] of every
// output quad q, the A operand of v_mfma_f32_4x4x1_16b_f32 with the lane's input value c as B (one lane per frame: the sixteen
// 4 x 4 blocks of the instruction are four frames each): ONE matrix instruction per input and output quad instead of two packed
// FMAs, one b128 read per four inputs instead of one per input, a quarter of the registers. Same sums in the same order as
// wr_mv / wr_mv_acc (bias, then the inputs in order: an fp32 MFMA is the fmaf chain). A layer's conv (one such matrix per tap),
// layer1x1 and head1x1 come this way (plan.h: wr_layer_layout; plan_wr.cpp packs them).
template <int OUT, int IN>
struct WrMatM
{
  static constexpr int Q = wr_pad4(OUT) / 4, I4 = wr_pad4(IN) / 4;
  f4 w[Q][I4];
  f4 b[Q];
};
template <int OUT, int IN>
__device__ __forceinline__ void wr_ld(WrMatM<OUT, IN>& m, const char* lds, unsigned wb, bool bias, unsigned bb)
{
  constexpr int Q = WrMatM<OUT, IN>::Q, I4 = WrMatM<OUT, IN>::I4;
  const unsigned cls_b = (threadIdx.x & 3u) * (unsigned)(Q * I4 * 16);
#pragma unroll
  for (int q = 0; q < Q; q++)
  {
#pragma unroll
    for (int c4 = 0; c4 < I4; c4++)
      m.w[q][c4] = lds_ld4(lds, wb + cls_b + (unsigned)((q * I4 + c4) * 16));
    m.b[q] = bias ? lds_ld4(lds, bb + (unsigned)q * 16u) : f4{0.f, 0.f, 0.f, 0.f};
  }
}
// acc (pairs) = b + W in
template <int OUT, int IN>
__device__ __forceinline__ void wr_mv(f2* acc, const float* in, const WrMatM<OUT, IN>& m)
{
  constexpr int Q = WrMatM<OUT, IN>::Q, P = wr_pairs(OUT);
  f4 a[Q];
#pragma unroll
  for (int q = 0; q < Q; q++)
    a[q] = m.b[q];
#pragma unroll
  for (int i = 0; i < IN; i++)
#pragma unroll
    for (int q = 0; q < Q; q++)
      a[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(m.w[q][i >> 2][i & 3], in[i], a[q], 0, 0, 0);
#pragma unroll
  for (int q = 0; q < Q; q++)
  {
    acc[2 * q] = f2{a[q][0], a[q][1]};
    if (2 * q + 1 < P)
      acc[2 * q + 1] = f2{a[q][2], a[q][3]};
  }
}
// acc (pairs) += W in  (no bias: the caller seeded acc)
template <int OUT, int IN>
__device__ __forceinline__ void wr_mv_acc(f2* acc, const float* in, const WrMatM<OUT, IN>& m)
{
  constexpr int Q = WrMatM<OUT, IN>::Q, P = wr_pairs(OUT);
  f4 a[Q];
#pragma unroll
  for (int q = 0; q < Q; q++)
  {
    const f2 lo = acc[2 * q], hi = 2 * q + 1 < P ? acc[2 * q + 1] : f2{0.f, 0.f};
    a[q] = f4{lo[0], lo[1], hi[0], hi[1]};
  }
#pragma unroll
  for (int i = 0; i < IN; i++)
#pragma unroll
    for (int q = 0; q < Q; q++)
      a[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(m.w[q][i >> 2][i & 3], in[i], a[q], 0, 0, 0);
#pragma unroll
  for (int q = 0; q < Q; q++)
  {
    acc[2 * q] = f2{a[q][0], a[q][1]};
    if (2 * q + 1 < P)
      acc[2 * q + 1] = f2{a[q][2], a[q][3]};
  }
}

// film.h:76-204 — v[d] = v[d] * scale[d] (+ shift[d]);  scale = Ws cond + bs, shift = Wh cond + bh
// block at `fb`: Ws, Wh, bs [pad4(D)], bh [pad4(D)]. The two matrices come in one of two forms (plan_wr.cpp: WrBuilder packs what
// wr_film_matrix_form says):
//   * vector form [COND][pad4(D)]: one broadcast b128 = four outputs' weights for one input = two v_pk_fma_f32;
//   * MATRIX form (a condition of 4 or 8 values — every FiLM of a model behind a condition_dsp): [lane % 4][pad4(D) / 4][COND] —
//     the lane's own row W[4 q + lane % 4][c] of every output quad q, the A operand of v_mfma_f32_4x4x1_16b_f32 with the
//     lane's condition value c as B (one lane per frame: the sixteen 4 x 4 blocks are four frames each) — ONE matrix
//     instruction per input and output quad instead of two packed FMAs, one b128 per four inputs instead of one per input, and
//     a quarter of the registers per matrix. Same sums in the same order (bias, then inputs 0 .. COND - 1; an fp32 MFMA is the
//     fmaf chain).
template <int D, int COND>
struct WrFilm
{
  static constexpr bool kM = wr_film_matrix_form(COND);
  static constexpr int Q = wr_pad4(D > 0 ? D : 1) / 4;
  WrMat<(D > 0 ? D : 1), (kM ? 1 : COND)> s, h; // vector form
  f4 ms[Q][kM ? COND / 4 : 1], mh[Q][kM ? COND / 4 : 1], mbs[Q], mbh[Q]; // matrix form: the lane's rows, the biases
};
template <int D, int COND>
__device__ __forceinline__ void wr_ld(WrFilm<D, COND>& f, const char* lds, unsigned fb, bool shift)
{
  if constexpr (D > 0)
  {
    constexpr unsigned kMat = (unsigned)COND * wr_pad4(D) * 4u, kVec = (unsigned)wr_pad4(D) * 4u;
    if constexpr (WrFilm<D, COND>::kM)
    {
      constexpr int Q = WrFilm<D, COND>::Q;
      const unsigned cls_b = (threadIdx.x & 3u) * (unsigned)(Q * COND * 4);
#pragma unroll
      for (int q = 0; q < Q; q++)
      {
#pragma unroll
        for (int c4 = 0; c4 < COND / 4; c4++)
          f.ms[q][c4] = lds_ld4(lds, fb + cls_b + (unsigned)((q * COND + c4 * 4) * 4));
        f.mbs[q] = lds_ld4(lds, fb + 2u * kMat + (unsigned)q * 16u);
      }
      if (shift)
      {
#pragma unroll
        for (int q = 0; q < Q; q++)
        {
#pragma unroll
          for (int c4 = 0; c4 < COND / 4; c4++)
            f.mh[q][c4] = lds_ld4(lds, fb + kMat + cls_b + (unsigned)((q * COND + c4 * 4) * 4));
          f.mbh[q] = lds_ld4(lds, fb + 2u * kMat + kVec + (unsigned)q * 16u);
        }
      }
    }
    else
    {
      wr_ld(f.s, lds, fb, true, fb + 2u * kMat);
      if (shift)
        wr_ld(f.h, lds, fb + kMat, true, fb + 2u * kMat + kVec);
    }
  }
}
template <int D, int COND>
__device__ __forceinline__ void wr_film(f2* v, const float* cond, const WrFilm<D, COND>& f, bool shift)
{
  if constexpr (D > 0)
  {
    if constexpr (WrFilm<D, COND>::kM)
    {
      constexpr int Q = WrFilm<D, COND>::Q, P = wr_pairs(D);
      f4 sc[Q];
#pragma unroll
      for (int q = 0; q < Q; q++)
        sc[q] = f.mbs[q];
#pragma unroll
      for (int c = 0; c < COND; c++)
#pragma unroll
        for (int q = 0; q < Q; q++)
          sc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(f.ms[q][c >> 2][c & 3], cond[c], sc[q], 0, 0, 0);
      if (shift)
      {
        f4 sh[Q];
#pragma unroll
        for (int q = 0; q < Q; q++)
          sh[q] = f.mbh[q];
#pragma unroll
        for (int c = 0; c < COND; c++)
#pragma unroll
          for (int q = 0; q < Q; q++)
            sh[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(f.mh[q][c >> 2][c & 3], cond[c], sh[q], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < Q; q++)
        {
          v[2 * q] = __builtin_elementwise_fma(v[2 * q], f2{sc[q][0], sc[q][1]}, f2{sh[q][0], sh[q][1]});
          if (2 * q + 1 < P)
            v[2 * q + 1] = __builtin_elementwise_fma(v[2 * q + 1], f2{sc[q][2], sc[q][3]}, f2{sh[q][2], sh[q][3]});
        }
      }
      else
      {
#pragma unroll
        for (int q = 0; q < Q; q++)
        {
          v[2 * q] *= f2{sc[q][0], sc[q][1]};
          if (2 * q + 1 < P)
            v[2 * q + 1] *= f2{sc[q][2], sc[q][3]};
        }
      }
    }
    else
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
// One activation value. The transcendental / dividing types take the hardware exp2 / rcp forms (device_common.h: mf::act_hw,
// abs error ~1e-7, a handful of instructions) instead of libm expf / tanhf and IEEE division (a dozen instructions each:
// a fifth of a wavenet_a2_max block); the piecewise-linear types are exact either way.
template <int T>
__device__ __forceinline__ float wr_act1(float x, float p0, float p1, float p2, float p3, float slope)
{
  if constexpr (T == ACT_TANH || T == ACT_FASTTANH || T == ACT_SIGMOID || T == ACT_SILU || T == ACT_SOFTSIGN || T == ACT_HARDSWISH)
    return mf::act_hw(T, x, p0);
  else if constexpr (T == ACT_FASTSIGMOID)
    return mf::fast_sigmoid_hw(x);
  else
    return d_act<T>(x, p0, p1, p2, p3, slope);
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
      v[(R0 + c) >> 1][(R0 + c) & 1] = wr_act1<T>(pget(v, R0 + c), p[0], p[1], p[2], p[3], ap.slope[c >> 2][c & 3]);
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
template <int COND, int C, int B, bool G, int K, int HO, int FM, int SM, int BL, int A1, int A2, int L1>
__device__ __forceinline__ void wr_layer(WrRegs& r, const WrOpS& op, char* lds, int lane, int posv)
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
  // the conv matrix [K * C][pad4(ZC)]: whole in registers while the taps arrive — or, for long kernels / wide layers
  // (more than 256 weights per lane), tap by tap, so that a per-model build never spills its way through a layer
  // the conv: one matrix-form matrix per tap; all of them in registers while the taps arrive — or, for long kernels on wide
  // layers (more than 256 floats of rows per lane), tap by tap, so that a per-model build never spills its way through a layer
  constexpr bool kConvByTap = K * wr_pad4(ZC) * wr_pad4(C) / 4 > 256;
  constexpr unsigned kTapB = (unsigned)(wr_pad4(ZC) * wr_pad4(C) * 4); // bytes of one tap's matrix
  WrMatM<ZC, C> m_conv[kConvByTap ? 1 : K];
  WrFilm<B, COND> f_apost;
  WrActP<G ? B : ZC> a_1;
  WrActP<B> a_2;
  WrMatM<C, B> m_l1;
  WrFilm<C, COND> f_l1;
  WrMatM<(HO > 0 ? HO : 1), B> m_h1;
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
  // the layer's ring: this lane's frame goes to index (position + lane) mod R, tap k is (K - 1 - k) * dilation behind it
  const int R = op.ring;
  int widx = __builtin_amdgcn_readlane(posv, op.slot) + lane; // (write positions: lane = slot)
  widx -= widx >= R ? R : 0;
  const unsigned hb = (unsigned)op.hist * 4u;
  wr_ring_put<C>(lds, hb, widx, R, ci);
  float taps[K * C]; // [k][i]: tap k looks (K - 1 - k) * dilation frames back (conv1d.cpp: the last tap is "now")
#pragma unroll
  for (int k = 0; k + 1 < K; k++)
  {
    int idx = widx - (K - 1 - k) * op.dil;
    idx += idx < 0 ? R : 0;
    wr_ring_get<C>(lds, hb, idx, R, taps + k * C);
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
#pragma unroll
  for (int k = 0; k < (kConvByTap ? 1 : K); k++) // (by tap: tap 0 and the bias)
    wr_ld(m_conv[k], lds, wb + L.conv * 4u + (unsigned)k * kTapB, k == 0, wb + L.conv_b * 4u);
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
  wr_mv(z, taps, m_conv[0]);
  if constexpr (!kConvByTap)
  {
#pragma unroll
    for (int k = 1; k < K; k++)
      wr_mv_acc(z, taps + k * C, m_conv[k]);
  }
  else
  {
#pragma unroll
    for (int k = 1; k < K; k++)
    {
      wr_ld(m_conv[0], lds, wb + L.conv * 4u + (unsigned)k * kTapB, false, 0u);
      wr_fence();
      wr_mv_acc(z, taps + k * C, m_conv[0]);
    }
  }
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
  if constexpr (L1 != 0)
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
  if (on(FILM_ACT_POST)) // (an odd B shares its last pair with gate row B: dead after the gating, its scale / shift columns are padding)
    wr_film<B, COND>(z, r.cond, f_apost, sh(FILM_ACT_POST));
  float zb[B]; // the B rows the 1x1s read
  wr_unpack<B>(zb, z);
  const bool l1_film = L1 != 0 && G && blended && on(FILM_LAYER1X1_POST); // quirk kept: only in the BLENDED branch, model.cpp:282-286
  if (l1_film)
    wr_ld(f_l1, lds, fb(FILM_LAYER1X1_POST), sh(FILM_LAYER1X1_POST));
  if (HO > 0 && on(FILM_HEAD1X1_POST))
    wr_ld(f_h1, lds, fb(FILM_HEAD1X1_POST), sh(FILM_HEAD1X1_POST));
  wr_fence();
  f2 l1[wr_pairs(C)];
  if constexpr (L1 != 0)
  {
    wr_mv(l1, zb, m_l1);
    if (l1_film)
      wr_film<C, COND>(l1, r.cond, f_l1, sh(FILM_LAYER1X1_POST));
  }

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
  // residual — model.cpp:354-392 (without a layer1x1 the layer's output is its input: :380-391)
  if constexpr (L1 != 0)
  {
#pragma unroll
    for (int i = 0; i < C; i++)
      r.x[i] += pget(l1, i);
  }
}

// WR_RUN: consecutive PLAIN layers (model.cpp:183-393 with no FiLM, no gating, no head1x1, condition size 1, kernel size
// 3, C = bottleneck <= 4, so every matrix row is one b128). One dispatch for the whole run; layer l + 1's weights and
// ring record are requested before layer l computes, so a layer costs one exposed LDS round trip (its taps) instead of
// four. Compact weight block (plan.h: wr_plain_layout), the same summation order as wr_layer.
// Round 6: the two matrices on v_mfma_f32_4x4x1_16b_f32, one lane per frame (the sixteen 4 x 4 blocks are four frames each): the
// lane holds ITS output row (lane % 4) of the conv and of the 1x1 — pad4(3 C) / 4 + 1 b128 reads per layer instead of 4 C —, the
// B operand is the lane's own tap / activation value: one matrix instruction per input instead of two packed FMAs. Same sums in
// the same order (bias, tap 0's channels, tap 1's, the current frame's; an fp32 MFMA is the fmaf chain).
template <int C>
struct WrPlainW
{
  f4 conv[wr_pad4(3 * C) / 4], conv_b, mix, l1, l1_b;
  i4 rec; // {-, ring area float offset, R, dilation | slot << 24}
};
template <int C, bool LDREC = true>
__device__ __forceinline__ void wr_plain_ld(WrPlainW<C>& w, const char* lds, unsigned wb, unsigned rec_b)
{
  constexpr WrPlainLayout L = wr_plain_layout(C); // the compact block of a plain layer (matrix form)
  if constexpr (LDREC) // (a program compiled in — NAM_WR_PROGRAMS — knows its records: constants)
    w.rec = *reinterpret_cast<const i4*>(lds + rec_b);
  constexpr int IN4 = wr_pad4(3 * C);
  const unsigned cls = threadIdx.x & 3u; // the lane's output row
#pragma unroll
  for (int q = 0; q < IN4 / 4; q++)
    w.conv[q] = lds_ld4(lds, wb + (unsigned)L.conv * 4u + cls * (unsigned)(IN4 * 4) + (unsigned)q * 16u);
  w.conv_b = lds_ld4(lds, wb + (unsigned)L.conv_b * 4u);
  w.mix = lds_ld4(lds, wb + (unsigned)L.mixin * 4u);
  w.l1 = lds_ld4(lds, wb + (unsigned)L.l1 * 4u + cls * 16u);
  w.l1_b = lds_ld4(lds, wb + (unsigned)L.l1_b * 4u);
}
// One plain layer with the weights in `w`; the next layer's weights / record (LDS byte addresses nxt_wb / nxt_rec) are
// requested BEHIND this layer's taps: LDS returns in order, so the arithmetic waits for the taps only (a counted
// lgkmcnt) while the 16 weight reads stream in under it.
template <int C, int ACT, bool LDREC = true>
__device__ __forceinline__ void wr_plain_layer(WrRegs& r, const WrPlainW<C>& w, WrPlainW<C>& nxt, unsigned nxt_wb,
                                               unsigned nxt_rec, char* lds, int lane, int posv, const i4 rec_k = i4{0, 0, 0, 0})
{
  const i4 rec = LDREC ? w.rec : rec_k;
  const int R = rec[2], dil = rec[3] & 0xffffff;
  int widx = __builtin_amdgcn_readlane(posv, LDREC ? __builtin_amdgcn_readfirstlane(rec[3] >> 24) : (rec[3] >> 24)) + lane;
  widx -= widx >= R ? R : 0;
  const unsigned hb = (unsigned)rec[1] * 4u;
  wr_ring_put<C>(lds, hb, widx, R, r.x);
  float t0[C], t1[C]; // taps 2 and 1 dilations back
  int i0 = widx - 2 * dil, i1 = widx - dil;
  i0 += i0 < 0 ? R : 0;
  i1 += i1 < 0 ? R : 0;
  wr_ring_get<C>(lds, hb, i0, R, t0);
  wr_ring_get<C>(lds, hb, i1, R, t1);
  wr_fence();
  wr_plain_ld<C, LDREC>(nxt, lds, nxt_wb, nxt_rec);
  wr_fence();
  const f4 m = w.mix * r.cond[0]; // input mixin (no bias): 0 + W cond
  f4 z = w.conv_b;
#pragma unroll
  for (int i = 0; i < C; i++)
    z = __builtin_amdgcn_mfma_f32_4x4x1f32(w.conv[i >> 2][i & 3], t0[i], z, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < C; i++)
    z = __builtin_amdgcn_mfma_f32_4x4x1f32(w.conv[(C + i) >> 2][(C + i) & 3], t1[i], z, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < C; i++)
    z = __builtin_amdgcn_mfma_f32_4x4x1f32(w.conv[(2 * C + i) >> 2][(2 * C + i) & 3], r.x[i], z, 0, 0, 0);
  z += m;
  float a[C];
#pragma unroll
  for (int c = 0; c < C; c++)
  {
    a[c] = wr_act1<ACT>(z[c], 0.f, 0.f, 0.f, 0.f, 0.f);
    r.hacc[c] += a[c];
  }
  f4 y = w.l1_b;
#pragma unroll
  for (int i = 0; i < C; i++)
    y = __builtin_amdgcn_mfma_f32_4x4x1f32(w.l1[i], a[i], y, 0, 0, 0);
#pragma unroll
  for (int c = 0; c < C; c++)
    r.x[c] += y[c];
}
template <int C, int ACT>
__device__ __forceinline__ void wr_run(WrRegs& r, const WrOpS& op, char* lds, int lane, int posv, int n_layers, int w_stride)
{
  WrPlainW<C> wa, wb;
  const unsigned w0 = (unsigned)op.w * 4u, ws = (unsigned)w_stride * 4u, rec0 = (unsigned)op.hist * 4u;
  wr_plain_ld(wa, lds, w0, rec0);
  // two layers per trip (register sets a / b swap roles); every load is unconditional (the last ones re-read the final
  // layer) so that the LDS counters stay exact
  for (int l = 0; l < n_layers; l += 2)
  {
    const unsigned l1 = (unsigned)min(l + 1, n_layers - 1), l2 = (unsigned)min(l + 2, n_layers - 1);
    wr_plain_layer<C, ACT>(r, wa, wb, w0 + l1 * ws, rec0 + l1 * 16u, lds, lane, posv);
    if (l + 1 < n_layers)
      wr_plain_layer<C, ACT>(r, wb, wa, w0 + l2 * ws, rec0 + l2 * 16u, lds, lane, posv);
  }
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

// head rechannel with taps (model.cpp:399-400, 547-548): a Conv1D(kernel KH, dilation op.dil) over the head accumulator,
// which therefore has a ring like a layer's conv input; W^T = [KH * HI][pad4(HS)], row = tap * HI + input
template <int HI, int HS, int KH>
__device__ __forceinline__ void wr_array_end_k(WrRegs& r, const WrOpS& op, char* lds, int lane, int posv)
{
  WrMat<HS, KH * HI> m;
  wr_ld(m, lds, (unsigned)op.w * 4u, true, (unsigned)op.w * 4u + (unsigned)(KH * HI * wr_pad4(HS)) * 4u); // (a zero bias when none)
  const int R = op.ring;
  int widx = __builtin_amdgcn_readlane(posv, op.slot) + lane;
  widx -= widx >= R ? R : 0;
  const unsigned hb = (unsigned)op.hist * 4u;
  wr_ring_put<HI>(lds, hb, widx, R, r.hacc);
  float taps[KH * HI];
#pragma unroll
  for (int k = 0; k + 1 < KH; k++)
  {
    int idx = widx - (KH - 1 - k) * op.dil;
    idx += idx < 0 ? R : 0;
    wr_ring_get<HI>(lds, hb, idx, R, taps + k * HI);
  }
#pragma unroll
  for (int i = 0; i < HI; i++)
    taps[(KH - 1) * HI + i] = r.hacc[i];
  f2 o[wr_pairs(HS)];
  wr_mv(o, taps, m);
  wr_unpack<HS>(r.hout, o);
}

// one layer of the post-stack head (detail::Head::process, model.cpp:86-103; built :21-44): the activation on the input — the last
// array's head output times head_scale for the first layer (:854-866), the previous layer's output after that —, then a
// Conv1D(kernel KH, dilation 1, bias) whose history is of the ACTIVATED input (the reference activates in place before
// Conv1D::Process). W^T = [KH * HI][pad4(HS)] | bias | activation parameters.
template <int HI, int HS, int KH, int ACT>
__device__ __forceinline__ void wr_post_head(WrRegs& r, const WrOpS& op, char* lds, int lane, int posv)
{
  WrMat<HS, KH * HI> m;
  const unsigned wb = (unsigned)op.w * 4u, bb = wb + (unsigned)(KH * HI * wr_pad4(HS)) * 4u;
  wr_ld(m, lds, wb, true, bb);
  WrActP<HI> ap;
  wr_ld(ap, lds, bb + (unsigned)wr_pad4(HS) * 4u);
  const float sc = op.scale();
  float v[kWrRegs];
#pragma unroll
  for (int i = 0; i < HI; i++)
    v[i] = wr_act1<ACT>(sc * r.hout[i], ap.p[0], ap.p[1], ap.p[2], ap.p[3], ap.slope[i >> 2][i & 3]);
  float taps[KH * HI];
  if constexpr (KH > 1)
  {
    const int R = op.ring;
    int widx = __builtin_amdgcn_readlane(posv, op.slot) + lane;
    widx -= widx >= R ? R : 0;
    const unsigned hb = (unsigned)op.hist * 4u;
    wr_ring_put<HI>(lds, hb, widx, R, v);
#pragma unroll
    for (int k = 0; k + 1 < KH; k++)
    {
      int idx = widx - (KH - 1 - k);
      idx += idx < 0 ? R : 0;
      wr_ring_get<HI>(lds, hb, idx, R, taps + k * HI);
    }
  }
#pragma unroll
  for (int i = 0; i < HI; i++)
    taps[(KH - 1) * HI + i] = v[i];
  f2 o[wr_pairs(HS)];
  wr_mv(o, taps, m);
  wr_unpack<HS>(r.hout, o);
}

// ---- A model's op PROGRAMS compiled in (round 6; NAM_WR_PROGRAMS, generated by plan_wr.cpp: WrShapeSet::header_text) ----
// The per-model build used to compile the model's layer SHAPES and still walk its program as data: every op cost a fetch
// (four LDS reads, thirteen v_readfirstlane), a two-level switch, and its offsets / ring length / dilation / slot / flags as
// scalar registers feeding address arithmetic — about a fifth of the instructions a lone wavefront issues per buffer (and a
// lone wavefront's time IS its instruction count). With the program in the header every op is a constant expression: the op
// loop is unrolled at compile time, the handler is chosen by `if constexpr`, every LDS address is an immediate.
#ifdef NAM_WR_PROGRAMS
// two forms of every program: as planned (CUT = 0: what one wavefront per stream runs) and with its WR_RUNs cut into sub-runs
// (CUT = 1: what two / four wavefronts per stream share — kWrProgSplit are the cuts of that form)
constexpr WrOpS kWrProgOps[2][NAM_WR_N_PROGRAMS][NAM_WR_MAX_OPS] = {NAM_WR_PROGRAM_OPS, NAM_WR_PROGRAM_OPS_CUT};
constexpr int kWrProgCount[2][NAM_WR_N_PROGRAMS] = {NAM_WR_PROGRAM_COUNTS, NAM_WR_PROGRAM_COUNTS_CUT};
constexpr int kWrProgSplit[NAM_WR_N_PROGRAMS][4] = NAM_WR_PROGRAM_SPLITS; // [0 .. 2]: four waves per stream, [3]: two (plan_wr.cpp: wr_program_cuts)
constexpr int kWrRunRecs[][4] = NAM_WR_RUN_RECS; // WR_RUN: {-, ring area float offset, R, dilation | slot << 24} per layer; op.slot = first row

template <int I0, int I1, class F>
__device__ __forceinline__ void wr_static_for(F&& f)
{
  if constexpr (I0 < I1)
  {
    f(std::integral_constant<int, I0>{});
    wr_static_for<I0 + 1, I1>(f);
  }
}
template <int ID>
__device__ __forceinline__ void wr_layer_by_id(WrRegs& r, const WrOpS& op, char* lds, int lane, int posv)
{
#define X(ID_, COND, C, B, G, K, HO, FM, SM, BL, A1, A2, L1) \
  if constexpr (ID == ID_) \
    wr_layer<COND, C, B, G, K, HO, FM, SM, BL, A1, A2, L1>(r, op, lds, lane, posv);
  WR_LAYER_SHAPES(X)
#undef X
}
// a WR_RUN with its layer count, weight stride and ring records as constants (wr_run's two-register-set rotation, unrolled)
template <int C, int ACT, int CUT, int P, int I>
__device__ __forceinline__ void wr_run_prog(WrRegs& r, char* lds, int lane, int posv)
{
  constexpr WrOpS op = kWrProgOps[CUT][P][I];
  constexpr int NL = op.n_in, R0 = op.slot;
  constexpr unsigned w0 = (unsigned)op.w * 4u, ws = (unsigned)op.n_out * 4u;
  WrPlainW<C> wa, wb;
  wr_plain_ld<C, false>(wa, lds, w0, 0u);
  wr_static_for<0, (NL + 1) / 2>([&](auto t_tag) {
    constexpr int l = 2 * decltype(t_tag)::value;
    constexpr unsigned l1 = (unsigned)(l + 1 < NL ? l + 1 : NL - 1), l2 = (unsigned)(l + 2 < NL ? l + 2 : NL - 1);
    wr_plain_layer<C, ACT, false>(r, wa, wb, w0 + l1 * ws, 0u, lds, lane, posv,
                                  i4{0, kWrRunRecs[R0 + l][1], kWrRunRecs[R0 + l][2], kWrRunRecs[R0 + l][3]});
    if constexpr (l + 1 < NL)
      wr_plain_layer<C, ACT, false>(r, wb, wa, w0 + l2 * ws, 0u, lds, lane, posv,
                                    i4{0, kWrRunRecs[R0 + l + 1][1], kWrRunRecs[R0 + l + 1][2], kWrRunRecs[R0 + l + 1][3]});
  });
}
template <int ID, int CUT, int P, int I>
__device__ __forceinline__ void wr_run_by_id(WrRegs& r, char* lds, int lane, int posv)
{
#define X(ID_, C, A) \
  if constexpr (ID == ID_) \
    wr_run_prog<C, A, CUT, P, I>(r, lds, lane, posv);
  WR_RUN_SHAPES(X)
#undef X
}
// op I of program P (everything but WR_OUTPUT, which needs the launch's windows: the caller's)
template <int CUT, int P, int I>
__device__ __forceinline__ void wr_exec_op(WrRegs& r, char* lds, int lane, int posv)
{
  constexpr WrOpS cur = kWrProgOps[CUT][P][I];
  if constexpr (cur.type == WR_LAYER)
    wr_layer_by_id<cur.shape>(r, cur, lds, lane, posv);
  else if constexpr (cur.type == WR_RUN)
    wr_run_by_id<cur.shape, CUT, P, I>(r, lds, lane, posv);
  else if constexpr (cur.type == WR_ARRAY_BEGIN)
  {
#define X(ID, IN, OUT) \
  if constexpr (cur.shape == ID) \
    wr_array_begin<IN, OUT>(r, cur, lds);
    WR_PAIR_SHAPES(X)
#undef X
  }
  else if constexpr (cur.type == WR_ARRAY_END)
  {
#define X(ID, IN, OUT) \
  if constexpr (cur.shape == ID) \
    wr_array_end<IN, OUT>(r, cur, lds);
    WR_PAIR_SHAPES(X)
#undef X
  }
  else if constexpr (cur.type == WR_ARRAY_END_K)
  {
#define X(ID, IN, OUT, KH) \
  if constexpr (cur.shape == ID) \
    wr_array_end_k<IN, OUT, KH>(r, cur, lds, lane, posv);
    WR_HEADK_SHAPES(X)
#undef X
  }
  else if constexpr (cur.type == WR_POST_HEAD)
  {
#define X(ID, IN, OUT, KH, ACT) \
  if constexpr (cur.shape == ID) \
    wr_post_head<IN, OUT, KH, ACT>(r, cur, lds, lane, posv);
    WR_POSTHEAD_SHAPES(X)
#undef X
  }
  else if constexpr (cur.type == WR_SET_COND)
  {
#pragma unroll
    for (int c = 0; c < kWrRegs; c++)
      r.cond[c] = cur.scale() * r.hout[c];
  }
}
#endif

} // namespace

// SET: which layer code the instantiation carries — 0: the fully described WR_LAYER shapes only (wavenet_a2_max: 344
// registers, no spills), 1: WR_RUN shapes only (plain stacks: 278 registers, a short dispatch), 2: everything (the
// run-time-flag layer shapes spill).
// NST = 2 / 4: the op program cut into NST parts (WrGroup::split_op, balanced on weights by the planner); wave s runs its
// part on buffer k - s while wave 0 works on buffer k — the pipeline of wave sets of kernel_a1_p4.hip for this kernel. A
// launch of N streams then keeps NST x N wavefronts busy instead of N: config 4's 512 streams fill the chip's 1,024
// SIMDs with two waves each, 256 streams with four. Every ring belongs to the wave that runs its layer; the registers
// (WrRegs: 40 floats per lane) travel through one-slot LDS queues with single-writer "produced" / "consumed" words
// polled from inline asm. Only launches that hold more than one buffer (sessions, renders, prewarm) are started this way.
template <int SET, int NST = 1>
__device__ __forceinline__ void wn_reg_body(const WrArgs& a)
{
  extern __shared__ __attribute__((aligned(16))) char lds_wr[];
  char* const lds = lds_wr;
  const int lane = (int)threadIdx.x & 63;
  const int S = NST > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) : 0; // stage (wave) of this thread
  // the width group of this workgroup (wavefront-uniform: everything below stays in scalar registers)
  int gi = 0;
#pragma unroll
  for (int k = 1; k < kWrMaxGroups; k++)
    if (k < a.n_groups && (int)blockIdx.x >= a.g[k].first)
      gi = k;
  const WrGroup& G = a.g[gi];
  const int member = (int)blockIdx.x - G.first;
  const int stream = G.stream_map ? G.stream_map[member] : member;
  // persistent session (persist_wave.h): blocks come from commands, not from a frame count
  const bool pers = a.ps.ring != nullptr;
  PersistWave pw;
  unsigned cmd_off = 0;
  float* const st = G.state + (long)stream * G.state_stride;
  int* const sti = reinterpret_cast<int*>(st);
  float* const st_ring = st + kWrPosInts;
  const int blob_floats = G.blob_floats, hist_floats = G.hist_floats;
  const unsigned ring_b = (unsigned)blob_floats * 4u; // LDS: [blob][rings][queues (NST > 1): queue q = stage q -> q + 1]
  const unsigned queue0_b = ring_b + (unsigned)hist_floats * 4u;
  constexpr unsigned kQRegsB = 5u * kWrRegs * 64u * 4u; // a queue: the registers | token [0..3] | produced [4] | consumed [5] | (queue 0) ready, first count [6, 7]
  int* const q0words = reinterpret_cast<int*>(lds + queue0_b + kQRegsB);
  unsigned done = 0; // NST > 1, last wave: commands finished
  if constexpr (NST == 1)
  {
    if (pers && !pw.begin(a.ps, (int)blockIdx.x, cmd_off))
    {
      pw.leave(a.ps, (int)blockIdx.x); // nothing to do
      return;
    }
  }
  else
  {
    // wave 0 looks for the launch's first command; its answer is the workgroup's
    if (S == 0)
    {
      const bool ready = !pers || pw.begin(a.ps, (int)blockIdx.x, cmd_off);
      if (lane < NST - 1) // (the queues' counters)
      {
        int* const qw = reinterpret_cast<int*>(lds + queue0_b + (unsigned)lane * (unsigned)kWrQueueBytes + kQRegsB);
        qw[4] = qw[5] = 0;
      }
      if (lane == 0)
      {
        q0words[6] = ready ? 1 : 0;
        q0words[7] = (int)pw.seq;
      }
    }
    __syncthreads();
    const bool ready = __builtin_amdgcn_readfirstlane(q0words[6]) != 0;
    done = (unsigned)__builtin_amdgcn_readfirstlane(q0words[7]);
    if (!ready)
    {
      if (S == 0)
        pw.leave(a.ps, (int)blockIdx.x); // nothing to do
      return;
    }
  }
  const bool whole = pers || a.n_frames > kBlock; // more than one block: the whole ring area comes in (and goes back)

  // Prologue. The write positions first (lane = slot): every ring address depends on them. A one-block launch then
  // requests the windows its taps reach — table entries {float offset, R, slot | gs << 8, o} straight from memory
  // (wavefront-uniform: scalar loads), lane j <-> ring index wrap(position - o + j) — before the weights, so that both
  // travel together; everything lands in LDS afterwards: [blob (weights, tables, program)][rings].
  const int pos_in = sti[lane];
  const int ring_len = lane < G.n_slots ? reinterpret_cast<const int*>(G.blob + G.tab_ring)[lane] : 0; // R of slot `lane`
  const i4* const tab_pf = reinterpret_cast<const i4*>(G.blob + G.tab_pf);
  const int n_pf = whole ? 0 : G.n_pf;
  constexpr int kWin = 64;
  float win[kWin];
  int win_off[kWin];
  auto window = [&](int e, float& v, int& off) {
    const i4 t = tab_pf[e];
    const int p = __builtin_amdgcn_readlane(pos_in, t[2] & 255);
    int idx = p - t[3] + lane;
    idx += idx < 0 ? t[1] : 0;
    idx -= idx >= t[1] ? t[1] : 0;
    off = t[0] + idx * (t[2] >> 8);
    v = st_ring[off];
  };
  if (n_pf > 0)
  {
#pragma unroll
    for (int u = 0; u < kWin; u++)
      window(min(u, n_pf - 1), win[u], win_off[u]);
  }
  const int t4 = (S * 64 + lane) * 4; // (NST > 1: every wave copies its share)
  for (int base = 0; base < blob_floats; base += 8 * 256 * NST)
  {
    f4 v[8];
#pragma unroll
    for (int u = 0; u < 8; u++)
    {
      const int i = base + u * 256 * NST + t4;
      v[u] = i < blob_floats ? *reinterpret_cast<const f4*>(G.blob + i) : f4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 8; u++)
    {
      const int i = base + u * 256 * NST + t4;
      if (i < blob_floats)
        lds_st4(lds, (unsigned)i * 4u, v[u]);
    }
  }
  if (whole)
  {
    for (int base = 0; base < hist_floats; base += 8 * 256 * NST)
    {
      f4 v[8];
#pragma unroll
      for (int u = 0; u < 8; u++)
      {
        const int i = base + u * 256 * NST + t4;
        v[u] = i < hist_floats ? *reinterpret_cast<const f4*>(st_ring + i) : f4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 8; u++)
      {
        const int i = base + u * 256 * NST + t4;
        if (i < hist_floats)
          lds_st4(lds, ring_b + (unsigned)i * 4u, v[u]);
      }
    }
  }
  else if (n_pf > 0)
  {
#pragma unroll
    for (int u = 0; u < kWin; u++)
      if (u < n_pf)
        lds_st1(lds, ring_b + (unsigned)win_off[u] * 4u, win[u]);
    // further windows: their table entries come from the LDS copy of the blob (no extra memory round trip)
    const unsigned tab_b = (unsigned)G.tab_pf * 4u;
    for (int e0 = kWin; e0 < n_pf; e0 += kWin)
    {
#pragma unroll
      for (int u = 0; u < kWin; u++)
      {
        const i4 t = *reinterpret_cast<const i4*>(lds + tab_b + (unsigned)min(e0 + u, n_pf - 1) * 16u);
        const int p = __builtin_amdgcn_readlane(pos_in, __builtin_amdgcn_readfirstlane(t[2] & 255));
        int idx = p - t[3] + lane;
        idx += idx < 0 ? t[1] : 0;
        idx -= idx >= t[1] ? t[1] : 0;
        win_off[u] = t[0] + idx * (t[2] >> 8);
        win[u] = st_ring[win_off[u]];
      }
#pragma unroll
      for (int u = 0; u < kWin; u++)
        if (e0 + u < n_pf)
          lds_st1(lds, ring_b + (unsigned)win_off[u] * 4u, win[u]);
    }
  }
  const float* const in = a.in ? a.in + (long)stream * a.in_ch * a.io_stride : nullptr;
  float* const out = a.out ? a.out + (long)stream * a.out_ch * a.io_stride : nullptr;
  const unsigned ops_b = (unsigned)G.tab_ops * 4u;
  const int n_ops = G.n_ops;
  int posv = pos_in; // the rings' write positions, lane = slot

  int n = kBlock;
  // The next block's input samples are requested half a block ahead (a memory round trip per block otherwise sits in
  // front of the first layer): the next window of a multi-block launch, or — persistent session — the window of the
  // next command when the early look at the ring already shows it.
  float in_pf[kWrRegs];
  int pf_off = -1; // the frame offset in_pf belongs to; -1 = none
  auto load_in = [&](float* dst, int off, int count) {
#pragma unroll
    for (int c = 0; c < kWrRegs; c++)
    {
      dst[c] = 0.0f;
      if (in && c < a.in_ch && lane < count)
        dst[c] = pers ? persist_in(in + (long)c * a.io_stride + off + lane) : in[(long)c * a.io_stride + off + lane];
    }
  };
  // NST > 1: this wave's share of the program (cuts at a quarter, a half, three quarters of the weights; two stages use the
  // middle one), and the queues between the waves (kernel_a1_p4.hip: wait_word, queue_put / take)
  int oi0 = 0, oi1 = n_ops;
  if constexpr (NST == 2)
  {
    const int c = max(1, min(G.split_op[1], n_ops - 1));
    oi0 = S == 1 ? c : 0;
    oi1 = S == 0 ? c : n_ops;
  }
  else if constexpr (NST == 4)
  {
    const int c0 = max(1, min(G.split_op[0], n_ops - 3)), c1 = max(c0 + 1, min(G.split_op[1], n_ops - 2)),
              c2 = max(c1 + 1, min(G.split_op[2], n_ops - 1));
    oi0 = S == 0 ? 0 : S == 1 ? c0 : S == 2 ? c1 : c2;
    oi1 = S == 0 ? c0 : S == 1 ? c1 : S == 2 ? c2 : n_ops;
  }
  auto wait_word = [&](unsigned byte_addr, int want) { // until the LDS word has reached `want`
    int tmp;
    asm volatile("1:\n\tds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_sub_u32 %0, %0, %2\n\tv_cmp_gt_i32 vcc, 0, %0\n\t"
                 "s_cbranch_vccz 2f\n\ts_sleep 1\n\ts_branch 1b\n2:"
                 : "=&v"(tmp)
                 : "v"(byte_addr), "v"(want)
                 : "vcc");
  };
  auto queue_put = [&](int qi, int k, const WrRegs& q, int f0_, int n_, bool is_exit) {
    const unsigned queue_b = queue0_b + (unsigned)qi * (unsigned)kWrQueueBytes, qw_b = queue_b + kQRegsB;
    int* const qwords = reinterpret_cast<int*>(lds + qw_b);
    wait_word(qw_b + 20u, k); // the slot is free once buffer k - 1 has been taken out of it
    asm volatile("" ::: "memory");
    if (!is_exit)
    {
#pragma unroll
      for (int c = 0; c < kWrRegs; c++)
      {
        lds_st1(lds, queue_b + (unsigned)((0 * kWrRegs + c) * 64 + lane) * 4u, q.in[c]);
        lds_st1(lds, queue_b + (unsigned)((1 * kWrRegs + c) * 64 + lane) * 4u, q.x[c]);
        lds_st1(lds, queue_b + (unsigned)((2 * kWrRegs + c) * 64 + lane) * 4u, q.cond[c]);
        lds_st1(lds, queue_b + (unsigned)((3 * kWrRegs + c) * 64 + lane) * 4u, q.hacc[c]);
        lds_st1(lds, queue_b + (unsigned)((4 * kWrRegs + c) * 64 + lane) * 4u, q.hout[c]);
      }
    }
    if (lane == 0)
    {
      qwords[0] = f0_;
      qwords[1] = n_;
      qwords[2] = is_exit ? 1 : 0;
    }
    asm volatile("" ::: "memory");
    if (lane == 0)
      __hip_atomic_store(qwords + 4, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
  };
  auto queue_take = [&](int qi, int k, WrRegs& q, int& f0_, int& n_, bool& is_exit) {
    const unsigned queue_b = queue0_b + (unsigned)qi * (unsigned)kWrQueueBytes, qw_b = queue_b + kQRegsB;
    int* const qwords = reinterpret_cast<int*>(lds + qw_b);
    wait_word(qw_b + 16u, k + 1);
    asm volatile("" ::: "memory");
    f0_ = __builtin_amdgcn_readfirstlane(qwords[0]);
    n_ = __builtin_amdgcn_readfirstlane(qwords[1]);
    is_exit = __builtin_amdgcn_readfirstlane(qwords[2]) != 0;
#pragma unroll
    for (int c = 0; c < kWrRegs; c++)
    {
      q.in[c] = lds_ld1(lds, queue_b + (unsigned)((0 * kWrRegs + c) * 64 + lane) * 4u);
      q.x[c] = lds_ld1(lds, queue_b + (unsigned)((1 * kWrRegs + c) * 64 + lane) * 4u);
      q.cond[c] = lds_ld1(lds, queue_b + (unsigned)((2 * kWrRegs + c) * 64 + lane) * 4u);
      q.hacc[c] = lds_ld1(lds, queue_b + (unsigned)((3 * kWrRegs + c) * 64 + lane) * 4u);
      q.hout[c] = lds_ld1(lds, queue_b + (unsigned)((4 * kWrRegs + c) * 64 + lane) * 4u);
    }
#pragma unroll
    for (int c = 0; c < kWrRegs; c++)
      asm volatile("" ::"v"(q.in[c]), "v"(q.x[c]), "v"(q.cond[c]), "v"(q.hacc[c]), "v"(q.hout[c]) : "memory"); // (in registers: the slot may be reused)
    if (lane == 0)
      __hip_atomic_store(qwords + 5, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
  };
  if constexpr (NST > 1)
    __syncthreads(); // the blob and the rings are in LDS (every wave copied its share)
  const int pf_at = (oi0 + oi1) >> 1;
  int kbuf = 0; // NST > 1: buffers this wave has handed over / taken
  for (int f0 = pers ? (int)cmd_off : 0;;)
  {
    WrRegs r;
    if (NST == 1 || S == 0)
    {
      n = pers ? kBlock : min(kBlock, a.n_frames - f0);
      if (pers)
        pw.look_ahead(a.ps);
      if (pf_off == f0)
      {
#pragma unroll
        for (int c = 0; c < kWrRegs; c++)
          r.in[c] = in_pf[c];
      }
      else
        load_in(r.in, f0, n);
      pf_off = -1;
#pragma unroll
      for (int c = 0; c < kWrRegs; c++)
      {
        r.cond[c] = r.in[c]; // a net without condition_dsp (and the nested net itself) is conditioned on its input
        r.x[c] = r.hacc[c] = r.hout[c] = 0.0f;
      }
    }
    else
    {
      bool is_exit = false;
      queue_take(S - 1, kbuf, r, f0, n, is_exit);
      if (is_exit)
      {
        if (S + 1 < NST)
          queue_put(S, kbuf, r, 0, 0, true); // pass the EXIT on
        break;
      }
    }
#ifdef NAM_WR_PROGRAMS
    // the group's program, compiled in: this wave's part of it, op by op
    auto run_program = [&](auto p_tag) {
      constexpr int P = decltype(p_tag)::value, CUT = NST > 1 ? 1 : 0;
      constexpr int NOPS = kWrProgCount[CUT][P];
      auto part = [&](auto i0_tag, auto i1_tag) {
        constexpr int I0 = decltype(i0_tag)::value, I1 = decltype(i1_tag)::value, PF = (I0 + I1) >> 1;
        wr_static_for<I0, I1>([&](auto i_tag) {
          constexpr int I = decltype(i_tag)::value;
          constexpr WrOpS cur = kWrProgOps[CUT][P][I];
#ifdef NAM_WR_MARKERS // (developer builds: tools/isa_regions.py --markers counts the instructions between these comments)
          asm volatile("; nam_op program %0 op %1 type %2 stages %3" ::"i"(P), "i"(I), "i"(cur.type), "i"(NST));
#endif
          if constexpr (I == PF)
          {
            if (NST == 1 || S == 0)
            {
              int nf = -1, nn = kBlock;
              if (pers)
              {
                if ((unsigned)(pw.spec >> 32) == pw.seq + 2u)
                  nf = (int)(unsigned)pw.spec;
              }
              else if (f0 + kBlock < a.n_frames)
              {
                nf = f0 + kBlock;
                nn = min(kBlock, a.n_frames - nf);
              }
              if (nf >= 0)
              {
                load_in(in_pf, nf, nn);
                pf_off = nf;
              }
            }
          }
          if constexpr (cur.type == WR_OUTPUT)
          {
            if (out)
            {
#pragma unroll
              for (int c = 0; c < kWrRegs; c++)
                if (c < cur.n_out && lane < n)
                  out[(long)c * a.io_stride + f0 + lane] = cur.scale() * r.hout[c];
            }
          }
          else
            wr_exec_op<CUT, P, I>(r, lds, lane, posv);
        });
      };
      using std::integral_constant;
      if constexpr (NST == 1)
        part(integral_constant<int, 0>{}, integral_constant<int, NOPS>{});
      else if constexpr (NST == 2)
      {
        constexpr int c = kWrProgSplit[P][3] < 1 ? 1 : kWrProgSplit[P][3] > NOPS - 1 ? NOPS - 1 : kWrProgSplit[P][3];
        if (S == 0)
          part(integral_constant<int, 0>{}, integral_constant<int, c>{});
        else
          part(integral_constant<int, c>{}, integral_constant<int, NOPS>{});
      }
      else
      {
        constexpr int s0 = kWrProgSplit[P][0], s1 = kWrProgSplit[P][1], s2 = kWrProgSplit[P][2];
        constexpr int c0 = s0 < 1 ? 1 : s0 > NOPS - 3 ? NOPS - 3 : s0;
        constexpr int c1 = (s1 > NOPS - 2 ? NOPS - 2 : s1) < c0 + 1 ? c0 + 1 : (s1 > NOPS - 2 ? NOPS - 2 : s1);
        constexpr int c2 = (s2 > NOPS - 1 ? NOPS - 1 : s2) < c1 + 1 ? c1 + 1 : (s2 > NOPS - 1 ? NOPS - 1 : s2);
        if (S == 0)
          part(integral_constant<int, 0>{}, integral_constant<int, c0>{});
        else if (S == 1)
          part(integral_constant<int, c0>{}, integral_constant<int, c1>{});
        else if (S == 2)
          part(integral_constant<int, c1>{}, integral_constant<int, c2>{});
        else
          part(integral_constant<int, c2>{}, integral_constant<int, NOPS>{});
      }
    };
    {
      bool ran = false;
      wr_static_for<0, NAM_WR_N_PROGRAMS>([&](auto p_tag) {
        if (!ran && G.prog == decltype(p_tag)::value)
        {
          run_program(p_tag);
          ran = true;
        }
      });
      if (!ran)
        __builtin_trap();
    }
#ifdef NAM_WR_MARKERS
    asm volatile("; nam_op program -1 op -1 type -1 stages %0" ::"i"(NST)); // (behind the program: the per-buffer remainder)
#endif
    (void)ops_b;
    (void)n_ops;
    (void)oi0;
    (void)oi1;
    (void)pf_at;
#else
    WrOpS cur = wr_fetch(lds, ops_b, oi0);
    for (int oi = oi0; oi < oi1; oi++)
    {
      const WrOpS nxt = wr_fetch(lds, ops_b, min(oi + 1, n_ops - 1)); // requested before this op runs
      if (oi == pf_at && (NST == 1 || S == 0))
      {
        int nf = -1, nn = kBlock;
        if (pers)
        {
          if ((unsigned)(pw.spec >> 32) == pw.seq + 2u)
            nf = (int)(unsigned)pw.spec;
        }
        else if (f0 + kBlock < a.n_frames)
        {
          nf = f0 + kBlock;
          nn = min(kBlock, a.n_frames - nf);
        }
        if (nf >= 0)
        {
          load_in(in_pf, nf, nn);
          pf_off = nf;
        }
      }
      switch (cur.type)
      {
        case WR_LAYER:
          if constexpr (SET != 1)
          {
            switch (cur.shape)
            {
#define X(ID, COND, C, B, G, K, HO, FM, SM, BL, A1, A2, L1) \
  case ID: \
    if constexpr (SET == 2 || FM >= 0) \
      wr_layer<COND, C, B, G, K, HO, FM, SM, BL, A1, A2, L1>(r, cur, lds, lane, posv); \
    else \
      __builtin_trap(); \
    break;
              WR_LAYER_SHAPES(X)
#undef X
              default: __builtin_trap();
            }
          }
          else
            __builtin_trap();
          break;
        case WR_RUN:
          if constexpr (SET != 0)
          {
            switch (cur.shape)
            {
#define X(ID, C, A) \
  case ID: wr_run<C, A>(r, cur, lds, lane, posv, cur.n_in, cur.n_out); break;
              WR_RUN_SHAPES(X)
#undef X
              default: __builtin_trap();
            }
          }
          else
            __builtin_trap();
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
        case WR_ARRAY_END_K:
          switch (cur.shape)
          {
#define X(ID, IN, OUT, KH) \
  case ID: wr_array_end_k<IN, OUT, KH>(r, cur, lds, lane, posv); break;
            WR_HEADK_SHAPES(X)
#undef X
            default: __builtin_trap();
          }
          break;
        case WR_POST_HEAD:
          switch (cur.shape)
          {
#define X(ID, IN, OUT, KH, ACT) \
  case ID: wr_post_head<IN, OUT, KH, ACT>(r, cur, lds, lane, posv); break;
            WR_POSTHEAD_SHAPES(X)
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
#endif
    // every ring moves on by the block's n frames (lane = slot; lanes without a slot stay at 0)
    posv += ring_len > 0 ? n : 0;
    posv -= posv >= ring_len ? ring_len : 0;
    if constexpr (NST > 1)
    {
      if (S + 1 < NST)
        queue_put(S, kbuf, r, f0, n, false);
      kbuf++;
      if (S > 0)
      {
        done++; // (the next buffer comes out of the queue)
        continue;
      }
    }
    bool more;
    if (pers)
    {
      more = pw.next(a.ps, (int)blockIdx.x, cmd_off); // false: ring empty, leave
      f0 = (int)cmd_off;
    }
    else
    {
      f0 += kBlock;
      more = f0 < a.n_frames;
    }
    if (!more)
    {
      if constexpr (NST > 1)
        queue_put(0, kbuf, r, 0, 0, true); // EXIT: the later waves finish what is in front of them and leave
      break;
    }
  }
  if constexpr (NST > 1)
    __syncthreads(); // every wave is done with its rings
  // the state goes back: everything, or the frames this launch's single block appended
  if (whole)
  {
    for (int base = 0; base < hist_floats; base += 4 * 256 * NST)
    {
#pragma unroll
      for (int u = 0; u < 4; u++)
      {
        const int i = base + u * 256 * NST + t4;
        if (i < hist_floats)
          *reinterpret_cast<f4*>(st_ring + i) = lds_ld4(lds, ring_b + (unsigned)i * 4u);
      }
    }
  }
  else
  {
    // `rows` table (from the LDS copy of the blob): one entry per channel; the n frames of this launch's block sit at
    // wrap(old position + lane)
    const unsigned tab_b = (unsigned)G.tab_rows * 4u;
    const int n_rows = G.n_rows;
    for (int e0 = 0; e0 < n_rows; e0 += 16)
    {
      float v[16];
      int off[16];
#pragma unroll
      for (int u = 0; u < 16; u++)
      {
        const i4 t = *reinterpret_cast<const i4*>(lds + tab_b + (unsigned)min(e0 + u, n_rows - 1) * 16u);
        int idx = __builtin_amdgcn_readlane(pos_in, __builtin_amdgcn_readfirstlane(t[2] & 255)) + lane;
        idx -= idx >= t[1] ? t[1] : 0;
        off[u] = t[0] + idx * (t[2] >> 8);
        v[u] = lds_ld1(lds, ring_b + (unsigned)off[u] * 4u);
      }
#pragma unroll
      for (int u = 0; u < 16; u++)
        if (e0 + u < n_rows && lane < n)
          st_ring[off[u]] = v[u];
    }
  }
  if constexpr (NST == 1)
  {
    sti[lane] = posv;
    if (pers)
      pw.leave(a.ps, (int)blockIdx.x);
  }
  else
  {
    if (S == NST - 1)
      sti[lane] = posv;
    if (pers)
    {
      // results visible, then the consumed-command count (PersistWave::leave): the last wave knows it; the other waves'
      // ring copies are ordered in front of the fence by the barrier below
      __syncthreads();
      if (S == NST - 1)
      {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        if (lane == 0)
        {
          a.ps.cons[blockIdx.x] = done;
          __hip_atomic_store(a.ps.done + blockIdx.x, done | 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
  }
}

#ifdef NAM_WR_JIT_SHAPES
// The per-model build (wr_jit.cpp): this file compiled with the model's own shape tables in front of it, one kernel
// holding exactly those shapes, found by name in the code object.
extern "C" __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void nam_wn_reg_jit(const WrArgs a)
{
  wn_reg_body<2>(a);
}
extern "C" __global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 1))) void nam_wn_reg_jit2(const WrArgs a)
{
  wn_reg_body<2, 2>(a); // two stages (two wavefronts per stream)
}
extern "C" __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void nam_wn_reg_jit4(const WrArgs a)
{
  wn_reg_body<2, 4>(a); // four stages
}
// The DENSE forms (round 6): the same bodies built for TWO wavefronts per SIMD (at most 256 vector registers). A lone wave issues
// one instruction every ~8 cycles whatever it is — half of what its SIMD could issue —, so a batch that leaves SIMDs idle or a
// model small enough to live in 256 registers runs two stages' waves side by side on a SIMD: the host uses a dense form only
// when the compiler fitted it without scratch (api_launch.cpp: wr_jit_function, launch_wr's duration model).
extern "C" __global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) void nam_wn_reg_jit2d(const WrArgs a)
{
  wn_reg_body<2, 2>(a);
}
extern "C" __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void nam_wn_reg_jit4d(const WrArgs a)
{
  wn_reg_body<2, 4>(a);
}
#else
template <int SET>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void nam_wn_reg_kernel(const WrArgs a)
{
  wn_reg_body<SET>(a);
}
template <int SET>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 1))) void nam_wn_reg2_kernel(const WrArgs a)
{
  wn_reg_body<SET, 2>(a); // two stages (two wavefronts per stream)
}
template <int SET>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void nam_wn_reg4_kernel(const WrArgs a)
{
  wn_reg_body<SET, 4>(a); // four stages
}

// a kernel compiled for one model's shapes (`fn`: hipFunction_t of nam_wn_reg_jit / nam_wn_reg_jit2 in that model's code object)
hipError_t launch_wn_reg_jit(void* fn, const WrArgs& a, int n_workgroups, int lds_bytes, int stages, hipStream_t stream)
{
  if (n_workgroups <= 0 || a.n_frames <= 0)
    return hipSuccess;
  WrArgs args = a;
  size_t size = sizeof(args);
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  if (tl_session_stop_event) // (kernels.h: nam_launch; the Ext form takes the grid in work-items)
    return hipExtModuleLaunchKernel(reinterpret_cast<hipFunction_t>(fn), (unsigned)n_workgroups * 64u * (unsigned)stages, 1, 1, 64u * (unsigned)stages, 1, 1,
                                    (size_t)lds_bytes, stream, nullptr, config, nullptr, tl_session_stop_event, 0);
  return hipModuleLaunchKernel(reinterpret_cast<hipFunction_t>(fn), (unsigned)n_workgroups, 1, 1, 64u * (unsigned)stages, 1, 1,
                               (unsigned)lds_bytes, stream, nullptr, config);
}

hipError_t launch_wn_reg(const WrArgs& a, int n_workgroups, int lds_bytes, bool layers, bool runs, bool rt_layers, int stages,
                         hipStream_t stream)
{
  if (n_workgroups <= 0 || a.n_frames <= 0)
    return hipSuccess;
  if (stages > 1)
  {
    static DynamicLdsLimit lds_limit2[6];
    auto launch2 = [&](auto kernel, int set) -> hipError_t {
      if (lds_bytes > 64 * 1024)
      {
        const hipError_t e = lds_limit2[set].ensure(reinterpret_cast<const void*>(kernel), kWrMaxLdsBytes);
        if (e != hipSuccess)
          return e;
      }
      nam_launch(kernel, dim3(n_workgroups), dim3(64 * stages), (unsigned)lds_bytes, stream, a);
      return hipGetLastError();
    };
    const int set = ((layers && runs) || rt_layers) ? 2 : runs ? 1 : 0;
    if (stages == 4)
      return set == 2 ? launch2(nam_wn_reg4_kernel<2>, 5) : set == 1 ? launch2(nam_wn_reg4_kernel<1>, 4) : launch2(nam_wn_reg4_kernel<0>, 3);
    return set == 2 ? launch2(nam_wn_reg2_kernel<2>, 2) : set == 1 ? launch2(nam_wn_reg2_kernel<1>, 1) : launch2(nam_wn_reg2_kernel<0>, 0);
  }
  // more than the default 64 KB of dynamic LDS per workgroup (long dilations at 4+ channels: the official nano size
  // keeps 68 KB of rings): raised once per instantiation
  static DynamicLdsLimit lds_limit[3]; // per instantiation, tracked per device (kernels.h)
  auto launch = [&](auto kernel, int set) -> hipError_t {
    if (lds_bytes > 64 * 1024)
    {
      const hipError_t e = lds_limit[set].ensure(reinterpret_cast<const void*>(kernel), kWrMaxLdsBytes);
      if (e != hipSuccess)
        return e;
    }
    nam_launch(kernel, dim3(n_workgroups), dim3(64), (unsigned)lds_bytes, stream, a);
    return hipGetLastError();
  };
  if ((layers && runs) || rt_layers)
    return launch(nam_wn_reg_kernel<2>, 2);
  if (runs)
    return launch(nam_wn_reg_kernel<1>, 1);
  return launch(nam_wn_reg_kernel<0>, 0);
}
#endif

} // namespace namhip
