// kernel_generic.hip — the op-program interpreter. File-level notes for all kernel_*.hip: hand-written HIP for gfx950 (MI355X, CDNA4). No CUDA path, no shims.
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
#include "device_common.h"

namespace namhip
{

// ------------------------------------------------------------------------------------------------
// Generic interpreter
// ------------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------------
// Where the time of this kernel goes: ONE wavefront per SIMD and nothing to switch to, so every instruction costs
// its issue slot and every dependent memory round trip is exposed. The first interpreter (round 1) spent 16 k scalar
// and 9 k vector instructions per 64-frame block of wavenet_a2_max for 2.3 k FMAs, with one LDS round trip per input
// channel and one HBM round trip per history tap and channel. This one is built around three rules:
//   * every tensor owns a multiple of FOUR rows and every weight matrix is zero-padded to match (plan_ops.cpp), so rows
//     are read, combined and written four at a time with compile-time trip counts — no per-row tests or clamps;
//   * an op's operands are all requested before the first one is used: a conv block costs one LDS round trip per
//     8 input channels (CI4 = 1 or 2 four-row chunks, CB = 4 .. 16 accumulators: template instances picked by a
//     switch), an elementwise op one per four rows;
//   * history is staged into LDS once per block by OP_STAGE (one HBM round trip for all rings).
// LDS is zero-filled at launch so that padding rows only ever hold finite values.
// ------------------------------------------------------------------------------------------------

// acc[0 .. CB) += W[rows ci0 .. ci0 + 4 CI4)[CB] . x, x = the four-row chunks at `xrow` (LDS float offsets, lane
// included), weights at float offset `wk` (row pitch cpad) in the LDS / global copy
template <int CI4, int CB, bool WLDS, class XF>
__device__ __forceinline__ void conv_chunk(float (&acc)[CB], const float* __restrict__ blob, const float* wlds, int wk,
                                           int cpad, XF&& xf)
{
  float x[4 * CI4];
#pragma unroll
  for (int u = 0; u < 4 * CI4; u++)
    x[u] = xf(u);
  float wv[4 * CI4][CB];
#pragma unroll
  for (int u = 0; u < 4 * CI4; u++)
  {
    if constexpr (WLDS)
    {
#pragma unroll
      for (int j = 0; j < CB; j += 4)
      {
        const mf_f4 q = *reinterpret_cast<const mf_f4*>(wlds + wk + u * cpad + j);
        wv[u][j] = q[0], wv[u][j + 1] = q[1], wv[u][j + 2] = q[2], wv[u][j + 3] = q[3];
      }
    }
    else
    {
#pragma unroll
      for (int j = 0; j < CB; j++)
        wv[u][j] = blob[wk + u * cpad + j];
    }
  }
#pragma unroll
  for (int u = 0; u < 4 * CI4; u++)
#pragma unroll
    for (int j = 0; j < CB; j++)
      acc[j] = fmaf(wv[u][j], x[u], acc[j]);
}

// One OP_CONV: dst[co][t] = (bias[co]) + sum_k sum_ci W[k][ci][co] * tap_k[ci][t]
// tap_k[ci][t] = src frame (t - L), L = (K-1-k)*dil: from the LDS block when t-L >= 0, else from the stream's
// history — staged in LDS by OP_STAGE (small lookbacks) or read from the ring in HBM. Afterwards the block is appended
// to the ring. CB = op.cb accumulators per output block (the whole conv when cout <= 16).
template <int CB, bool WLDS>
__device__ __forceinline__ void op_conv(const NamOp& op, float* lds, const float* __restrict__ blob, const float* wlds,
                                        float* st, int* wpos_tbl, const int lane, const int nvalid)
{
  const int cin = op.cin, cout = op.cout, cpad = op.cout_pad, K = op.k;
  const int cin_pad = (cin + 3) & ~3;
  const bool has_ring = op.state >= 0;
  const bool staged = has_ring && (op.flag & 4);
  const int film = has_ring ? 0 : op.flag; // (a conv with taps never carries a FiLM epilogue)
  const int R = op.ring;
  const int lookback = (K - 1) * op.dil;
  const float* src = lds + op.src + lane;
  const float* hist = lds + op.aux; // staged: [lookback][cin]
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
    if (op.b >= 0)
    {
#pragma unroll
      for (int j = 0; j < CB; j++)
        acc[j] = WLDS ? wlds[op.b + co0 + j] : blob[op.b + co0 + j];
    }
    else
    {
#pragma unroll
      for (int j = 0; j < CB; j++)
        acc[j] = 0.0f;
    }
    for (int k = 0; k < K; k++)
    {
      const int L = (K - 1 - k) * op.dil;
      const int wk0 = op.w + k * cin_pad * cpad + co0;
      if (L == 0)
      {
        // current frame: rows straight from the block (eight input channels per round trip, then four)
        int ci0 = 0;
        for (; ci0 + 8 <= cin_pad; ci0 += 8)
          conv_chunk<2, CB, WLDS>(acc, blob, wlds, wk0 + ci0 * cpad, cpad, [&](int u) { return src[(ci0 + u) * kBlock]; });
        if (ci0 < cin_pad)
          conv_chunk<1, CB, WLDS>(acc, blob, wlds, wk0 + ci0 * cpad, cpad, [&](int u) { return src[(ci0 + u) * kBlock]; });
      }
      else
      {
        const int tl = lane - L;
        const bool in_block = tl >= 0;
        const float* srow = lds + op.src + (in_block ? tl : 0);
        if (staged)
        {
          const float* hrow = hist + (in_block ? 0 : (lookback + tl) * cin);
          for (int ci0 = 0; ci0 < cin_pad; ci0 += 4)
            conv_chunk<1, CB, WLDS>(acc, blob, wlds, wk0 + ci0 * cpad, cpad, [&](int u) {
              const float xl = srow[(ci0 + u) * kBlock];
              const float xh = hrow[ci0 + u]; // (past the last channel: the next frame's row or the next history, finite)
              return in_block ? xl : xh;
            });
        }
        else
        {
          int ridx = wp + tl; // ring row of frame t - L (only used by lanes before the block)
          if (ridx < 0)
            ridx += R;
          if (in_block)
            ridx = 0; // keep the masked-off address in range
          for (int ci0 = 0; ci0 < cin_pad; ci0 += 4)
            conv_chunk<1, CB, WLDS>(acc, blob, wlds, wk0 + ci0 * cpad, cpad, [&](int u) {
              const float xl = srow[(ci0 + u) * kBlock];
              const float xr = ring[(size_t)ridx * cin + min(ci0 + u, cin - 1)];
              return in_block ? xl : xr;
            });
        }
      }
    }
    if (film)
    {
      // FiLM epilogue (film.h:76-204): the block holds the scales of `per` channels, then (flag 2) their shifts
      constexpr int PER_S = CB / 2;
      const int c0 = film == 2 ? co0 / 2 : co0;
      const float* xs = lds + op.aux + c0 * kBlock + lane;
      float* d = lds + op.dst + c0 * kBlock + lane;
      const int rows_left = ((cout + 3) & ~3) - c0; // (a FiLM over more than 8 / 16 channels: the last block is partial)
      if (film == 2)
      {
        float tv[PER_S];
#pragma unroll
        for (int j = 0; j < PER_S; j++)
          tv[j] = xs[j * kBlock];
#pragma unroll
        for (int j = 0; j < PER_S; j++)
          if (PER_S <= 4 || j < rows_left)
            d[j * kBlock] = fmaf(tv[j], acc[j], acc[j + PER_S]);
      }
      else
      {
        float tv[CB];
#pragma unroll
        for (int j = 0; j < CB; j++)
          tv[j] = xs[j * kBlock];
#pragma unroll
        for (int j = 0; j < CB; j++)
          if (CB <= 4 || j < rows_left)
            d[j * kBlock] = tv[j] * acc[j];
      }
    }
    else
    {
      float* d = lds + op.dst + co0 * kBlock + lane;
      if (co0 + CB <= ((cout + 3) & ~3))
      {
#pragma unroll
        for (int j = 0; j < CB; j++)
          d[j * kBlock] = acc[j];
      }
      else // (only the last block of a conv with more than 16 outputs)
      {
#pragma unroll
        for (int j = 0; j < CB; j++)
          if (co0 + j < cout)
            d[j * kBlock] = acc[j];
      }
    }
  }
  if (has_ring)
  {
    int widx = wp + lane;
    if (widx >= R)
      widx -= R;
    for (int ci0 = 0; ci0 < cin; ci0 += 4)
    {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; u++)
        v[u] = src[(ci0 + u) * kBlock];
      if (lane < nvalid)
      {
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (ci0 + u < cin)
            ring[(size_t)widx * cin + ci0 + u] = v[u];
      }
    }
    int nwp = wp + nvalid;
    if (nwp >= R)
      nwp -= R;
    if (lane == 0)
      wpos_tbl[op.ring_id] = nwp;
  }
}

// Elementwise ops over `n` channel rows (their tensors own pad4(n) rows), four rows per round trip: f(c, v[]) gets the
// row values of every operand and returns the value of dst row c.
template <int NSRC, class F>
__device__ __forceinline__ void rows4(float* lds, int dst, const int (&srcs)[NSRC], int n, int lane, F&& f)
{
  for (int c0 = 0; c0 < n; c0 += 4)
  {
    float v[NSRC][4];
#pragma unroll
    for (int s = 0; s < NSRC; s++)
#pragma unroll
      for (int u = 0; u < 4; u++)
        v[s][u] = lds[srcs[s] + (c0 + u) * kBlock + lane];
    float r[4];
#pragma unroll
    for (int u = 0; u < 4; u++)
    {
      float a[NSRC];
#pragma unroll
      for (int s = 0; s < NSRC; s++)
        a[s] = v[s][u];
      r[u] = f(c0 + u, a);
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
      lds[dst + (c0 + u) * kBlock + lane] = r[u];
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
  // activation rows + history area: zero, so that padding rows (never written by LOAD_IN / exact-size stores) are finite
  for (int i = lane; i < a.w_lds_off; i += 64)
    lds[i] = 0.0f;
  const int stream = a.stream_map ? a.stream_map[blockIdx.x] : (int)blockIdx.x;
  float* st = a.state + (size_t)stream * a.state_stride;
  int* wpos_tbl = reinterpret_cast<int*>(st);
  const float* in = a.in ? a.in + (size_t)stream * a.in_ch * a.io_stride : nullptr;
  float* out = a.out ? a.out + (size_t)stream * a.out_ch * a.io_stride : nullptr;

  for (int f0 = 0; f0 < a.n_frames; f0 += kBlock)
  {
    const int nvalid = min(kBlock, a.n_frames - f0);
    // the descriptor of the op after this one is requested before this op runs (the program ends with OP_END and
    // plan_ops.cpp pads it with one more so that pc + 1 is always readable)
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
        case OP_STAGE:
        {
          // a run of op.cout staging ops (this one first): request every ring's last `lookback` frames, then write
          // them to LDS — one memory round trip for all of them. Frame-major in both places: element e of the
          // history is float ((wp - lookback) * cin + e) mod (R * cin) of the ring.
          constexpr int kMax = 16; // plan_ops.cpp: Builder::kMaxStages
          const int n = op.cout;
          float v[kMax];
#pragma unroll
          for (int i = 0; i < kMax; i++)
            if (i < n)
            {
              const NamOp so = ops[pc + i];
              const int wp = uni(wpos_tbl[so.ring_id]);
              const int span = so.ring * so.cin;
              int e = (wp - so.k) * so.cin + lane;
              if (e < 0)
                e += span;
              if (e >= span)
                e -= span;
              v[i] = lane < so.cin * so.k ? st[so.state + e] : 0.0f;
            }
#pragma unroll
          for (int i = 0; i < kMax; i++)
            if (i < n)
            {
              const NamOp so = ops[pc + i];
              if (lane < so.cin * so.k)
                lds[so.dst + lane] = v[i];
            }
          pc += n - 1;
          next_op = ops[pc + 1];
          break;
        }
        case OP_STORE_OUT:
          if (out && lane < nvalid)
            for (int c = 0; c < op.cin; c++)
              out[(size_t)c * a.io_stride + f0 + lane] = lds[op.src + c * kBlock + lane];
          break;
        case OP_CONV:
          switch (op.cb)
          {
            case 4: op_conv<4, WLDS>(op, lds, blob, wlds, st, wpos_tbl, lane, nvalid); break;
            case 8: op_conv<8, WLDS>(op, lds, blob, wlds, st, wpos_tbl, lane, nvalid); break;
            case 12: op_conv<12, WLDS>(op, lds, blob, wlds, st, wpos_tbl, lane, nvalid); break;
            default: op_conv<16, WLDS>(op, lds, blob, wlds, st, wpos_tbl, lane, nvalid); break;
          }
          break;
        case OP_FILM:
        {
          const int srcs[3] = {op.src, op.aux, op.aux + (op.flag ? op.cout * kBlock : 0)};
          const bool shift = op.flag != 0;
          rows4<3>(lds, op.dst, srcs, op.cout, lane, [&](int, const float(&v)[3]) { return shift ? fmaf(v[0], v[1], v[2]) : v[0] * v[1]; });
          break;
        }
        case OP_ACT:
        {
          const float p0 = blob[op.w], p1 = blob[op.w + 1], p2 = blob[op.w + 2], p3 = blob[op.w + 3];
          const int ns = op.ring;
          const int srcs[1] = {op.dst};
          // one dispatch per op, not per channel row (the compare cascade of d_act_rt is ~25 scalar instructions)
          auto rows = [&](auto type_tag) {
            constexpr int T = decltype(type_tag)::value;
            rows4<1>(lds, op.dst, srcs, op.cout, lane, [&](int c, const float(&v)[1]) {
              float slope = 0.0f;
              if constexpr (T == ACT_PRELU)
              {
                // Activation::apply(float*, size) on column-major data: slopes[pos % n] (activations.h:283-297)
                const long pos = (long)(f0 + lane) * op.cout + c;
                slope = blob[op.w + 4 + (int)(pos % ns)];
              }
              return d_act<T>(v[0], p0, p1, p2, p3, slope);
            });
          };
#define NAM_ACT_ROWS(T) \
  case T: rows(std::integral_constant<int, T>{}); break;
          switch (op.k)
          {
            NAM_ACT_ROWS(ACT_TANH)
            NAM_ACT_ROWS(ACT_HARDTANH)
            NAM_ACT_ROWS(ACT_FASTTANH)
            NAM_ACT_ROWS(ACT_RELU)
            NAM_ACT_ROWS(ACT_LEAKYRELU)
            NAM_ACT_ROWS(ACT_PRELU)
            NAM_ACT_ROWS(ACT_SIGMOID)
            NAM_ACT_ROWS(ACT_SILU)
            NAM_ACT_ROWS(ACT_HARDSWISH)
            NAM_ACT_ROWS(ACT_LEAKYHARDTANH)
            NAM_ACT_ROWS(ACT_SOFTSIGN)
            NAM_ACT_ROWS(ACT_FASTSIGMOID)
            case ACT_LUT: // FastLUTActivation: table behind the four parameters (device_common.h: d_lut)
              rows4<1>(lds, op.dst, srcs, op.cout, lane, [&](int, const float(&v)[1]) { return d_lut(blob + op.w, v[0]); });
              break;
            case ACT_IDENTITY: break;
            default: __builtin_trap(); break;
          }
#undef NAM_ACT_ROWS
          break;
        }
        case OP_GATE:
        {
          // gating_activations.h:59-114 (gated) / :165-228 (blended); result in the top B rows
          const int B = op.cout;
          const float a0 = blob[op.w], a1 = blob[op.w + 1], a2 = blob[op.w + 2], a3 = blob[op.w + 3];
          const float g0 = blob[op.b], g1 = blob[op.b + 1], g2 = blob[op.b + 2], g3 = blob[op.b + 3];
          const int srcs[2] = {op.dst, op.dst + B * kBlock};
          rows4<2>(lds, op.dst, srcs, B, lane, [&](int c, const float(&v)[2]) {
            const float pre = v[0], gin = v[1];
            const float s1 = (op.k == ACT_PRELU) ? blob[op.w + 4 + c % op.ring] : 0.0f;
            const float s2 = (op.dil == ACT_PRELU) ? blob[op.b + 4 + c % op.ring_id] : 0.0f;
            const float av = op.k == ACT_LUT ? d_lut(blob + op.w, pre) : d_act_rt(op.k, pre, a0, a1, a2, a3, s1);
            const float gv = op.dil == ACT_LUT ? d_lut(blob + op.b, gin) : d_act_rt(op.dil, gin, g0, g1, g2, g3, s2);
            return (op.flag == GATING_GATED) ? av * gv : gv * av + (1.0f - gv) * pre;
          });
          break;
        }
        case OP_ADD:
        {
          const int srcs[2] = {op.src, op.aux};
          rows4<2>(lds, op.dst, srcs, op.cout, lane, [&](int, const float(&v)[2]) { return v[0] + v[1]; });
          break;
        }
        case OP_COPY:
        {
          const int srcs[1] = {op.src};
          rows4<1>(lds, op.dst, srcs, op.cout, lane, [&](int, const float(&v)[1]) { return v[0]; });
          break;
        }
        case OP_ZERO:
          for (int c = 0; c < op.cout; c++)
            lds[op.dst + c * kBlock + lane] = 0.0f;
          break;
        case OP_SCALE:
        {
          const float sc = blob[op.w];
          const int srcs[1] = {op.src};
          rows4<1>(lds, op.dst, srcs, op.cout, lane, [&](int, const float(&v)[1]) { return sc * v[0]; });
          break;
        }
        default: break;
      }
      // One wavefront per workgroup: LDS operations of a wavefront execute in order, so the next op sees this op's rows
      // without a barrier or a wait; the optimiser only has to keep the accesses in program order.
      asm volatile("" ::: "memory");
    }
  }
}

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

} // namespace namhip
