// kernel_lstm.hip — the LSTM kernels (lanes = streams; 16 streams per wavefront on the matrix cores) and the
// per-stream state fill.
#include "device_common.h"
#include "il_common.h"
#include "persist_wave.h"

namespace namhip
{

// ------------------------------------------------------------------------------------------------
// LSTM: lanes = streams (a true per-sample recurrence — NAM/lstm.cpp:103-168)
// ------------------------------------------------------------------------------------------------
// GS: the per-lane columns (h, c, gate pre-activations) live in a global-memory scratch area instead of LDS — the cell is
// too large for a CU's 160 KB (the reference has no size limit, NAM/lstm.cpp:31-68); only the I/O tiles stay in LDS.
// Each lane touches its own column only ([row][64 lanes]: coalesced 256-byte rows).
template <bool GS>
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
  float* hs = GS ? a.scratch + (size_t)blockIdx.x * (size_t)(2 * NL + 4) * H * kBlock : tile_out + out_ch * kBlock * 65; // [NL][H][64]
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
template <int NL, int NT, bool FAST>
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
          if constexpr (FAST)
          {
            // fast_sigmoid(x) = 0.5 (fast_tanh(x / 2) + 1): the four gates are one 4-wide fast_tanh (same operations
            // per element as the scalar forms, so the same bits; the compiler pairs them into packed instructions)
            const f4 q = {0.5f * acc[T][0], 0.5f * acc[T][1], acc[T][2], 0.5f * acc[T][3]};
            f4 t;
#pragma unroll
            for (int e = 0; e < 4; e++)
              t[e] = mf::fast_tanh_hw(q[e]);
            cn = (0.5f * (t[1] + 1.0f)) * c[l][T] + (0.5f * (t[0] + 1.0f)) * t[2];
            hv = (0.5f * (t[3] + 1.0f)) * mf::fast_tanh_hw(cn);
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
// Small LSTMs (hidden <= 4: lstm.nam has 3 units), one GATE ROW PER LANE: a stream is a 16-lane DPP row, lane
// q = 4 u + k of the row computes gate k (i, f, g, o) of hidden unit u — one dot product of length I + H and ONE
// activation per lane and time step instead of a whole unit's 16 MACs + 5 activations (the 16-streams-per-wavefront
// MFMA kernel above: ~110 instructions per step on one wave; with a lone wave per SIMD the instruction count IS the
// time). The four gates of a unit meet through quad_perm DPP moves (c and h are replicated in the quad), the
// hidden state goes back to all 16 lanes through row_newbcast DPP moves; sigmoid = 0.5 tanh(x / 2) + 0.5 lets every
// lane run the same activation code with per-lane constants. 4 streams per wavefront, one wavefront per workgroup:
// 1,024 streams = 256 workgroups = every CU (the MFMA kernel keeps 64 wavefronts busy). I/O tiles go through LDS for
// coalescing exactly as in the other LSTM kernels; state layout [layer][h | c][H] is shared with them.
// Reference: NAM/lstm.cpp:31-68 (cell), :103-168 (process), gate order i, f, g, o; fast forms :48-58.
// ------------------------------------------------------------------------------------------------
namespace lrow
{
template <int N>
__device__ __forceinline__ float row_bcast(float v) // lane N of the 16-lane row -> every lane of the row
{
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + N, 0xf, 0xf, true));
}
template <int K>
__device__ __forceinline__ float quad_bcast(float v) // lane K of the quad -> every lane of the quad
{
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), K * 0x55, 0xf, 0xf, true));
}
template <bool FAST>
__device__ __forceinline__ float tanh_like(float x)
{
  return FAST ? mf::fast_tanh_hw(x) : mf::tanh_hw(x);
}
// The step in as few instructions as it takes: a lone wave issues one every ~8.6 cycles whatever it is, so the instruction
// COUNT of the step is its time (profiles/r04/valu_rate_microbench.txt).
// tanh_like(x) = x * ratio(x). FAST: the reference's rational (activations.h:29-41)
//   x (a + a |x| + (b + c |x|) x^2) / (d + (d + x^2) |x + e x |x||)
// with |x + e x |x|| = |x| (1 + e |x|) (e > 0), a gate's factor A (1 or 0.5) folded into the numerator's coefficients
// (per-lane registers kn, kc) and the independent pairs on v_pk_fma_f32: (x^2 + d, 1 + e |x|) and the numerator's two
// linear factors. `late`: a value from the end of the sequence (fake dependencies).
using f2 = __attribute__((ext_vector_type(2))) float;
template <bool FAST>
__device__ __forceinline__ float ratio(const float x, const float A, const f2 kn, const f2 kc, float& late)
{
  if constexpr (FAST)
  {
    // with q = x^2 + d: numerator / x = t1 x^2 + t2 = t1 q + (t2 - d t1), and t2 - d t1 is linear in |x| like t2: the x^2 never
    // has to exist by itself
    const float ax = __builtin_fabsf(x);
    const f2 axax = {ax, ax};
    const f2 qw = __builtin_elementwise_fma(axax, f2{ax, 0.814642734961073f}, f2{2.44506634652299f, 1.0f}); // (x^2 + d, 1 + e |x|)
    const f2 t = __builtin_elementwise_fma(kn, axax, kc); // A (b + c |x|), A ((a - d b) + (a - d c) |x|)
    const float z = ax * qw.y;
    const float den = __builtin_fmaf(qw.x, z, 2.44506634652299f);
    late = den;
    const float n = __builtin_fmaf(t.x, qw.x, t.y);
    return n * mf::rcp(den);
  }
  else
  {
    // A tanh(x) / x is not what the exp form gives: the caller multiplies by x again — keep the exact form: A tanh(x) = x * (A tanh(x) / x)
    // would divide by zero, so the libm form hands back A tanh(x) and sets `late` to 0 (see the callers)
    const float t = A * mf::tanh_hw(x);
    late = t;
    return t;
  }
}
// acc += sum over the NH hidden units of (lane 4 j of the 16-lane row of h) * w[j]: one VOP2 DPP instruction per term (the
// compiler's own form is a DPP move + an FMA each: its DPP combiner runs before v_fma becomes v_fmac), ONE asm statement (the
// compiler pads every asm statement with a wait state of its own). Two wait states in front: a DPP read needs them between
// the source's write (the step before) and itself, and the hazard recogniser does not look into inline asm.
#define NAM_LROW_TERM(W, N) "v_fmac_f32_dpp %0, %1, " W " row_newbcast:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1"
template <int NH>
__device__ __forceinline__ void recurrent_terms(float& acc, const float h, const float* w)
{
  if constexpr (NH == 1)
    asm("s_nop 1\n\t" NAM_LROW_TERM("%2", 0) : "+v"(acc) : "v"(h), "v"(w[0]));
  else if constexpr (NH == 2)
    asm("s_nop 1\n\t" NAM_LROW_TERM("%2", 0) "\n\t" NAM_LROW_TERM("%3", 4) : "+v"(acc) : "v"(h), "v"(w[0]), "v"(w[1]));
  else if constexpr (NH == 3)
    asm("s_nop 1\n\t" NAM_LROW_TERM("%2", 0) "\n\t" NAM_LROW_TERM("%3", 4) "\n\t" NAM_LROW_TERM("%4", 8)
        : "+v"(acc)
        : "v"(h), "v"(w[0]), "v"(w[1]), "v"(w[2]));
  else
    asm("s_nop 1\n\t" NAM_LROW_TERM("%2", 0) "\n\t" NAM_LROW_TERM("%3", 4) "\n\t" NAM_LROW_TERM("%4", 8) "\n\t" NAM_LROW_TERM("%5", 12)
        : "+v"(acc)
        : "v"(h), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]));
}
// The whole pre-activation of layer 0's gate row: bias + input term(s) + recurrent terms in ONE asm statement — the input term
// (a DPP move of the sample's lane + an FMA with the bias; independent of h) stands in front and IS the two wait states the
// first DPP read of h needs. Operands: %0 acc, %1 scratch, %2 x0, %3 wi0, %4 wb, %5 x1, %6 wi1, %7 h, %8 .. %11 wh, %12 = lane of
// the sample in its 16-lane row.
#define NAM_LROW_DPP " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define NAM_LROW_X0 "v_mov_b32_dpp %1, %2 row_newbcast:%12" NAM_LROW_DPP "v_fma_f32 %0, %3, %1, %4\n\t"
#define NAM_LROW_X1 "v_fmac_f32_dpp %0, %5, %6 row_newbcast:%12" NAM_LROW_DPP
#define NAM_LROW_H0 "v_fmac_f32_dpp %0, %7, %8 row_newbcast:0" NAM_LROW_DPP
#define NAM_LROW_H1 "v_fmac_f32_dpp %0, %7, %9 row_newbcast:4" NAM_LROW_DPP
#define NAM_LROW_H2 "v_fmac_f32_dpp %0, %7, %10 row_newbcast:8" NAM_LROW_DPP
#define NAM_LROW_H3 "v_fmac_f32_dpp %0, %7, %11 row_newbcast:12" NAM_LROW_DPP
#define NAM_LROW_ASM(TEXT) \
  asm(TEXT : "=&v"(acc), "=&v"(tmp) \
      : "v"(x0), "v"(wi[0]), "v"(wb), "v"(x1), "v"(wi[NI > 1 ? 1 : 0]), "v"(h), "v"(wh[0]), "v"(wh[NH > 1 ? 1 : 0]), "v"(wh[NH > 2 ? 2 : 0]), \
        "v"(wh[NH > 3 ? 3 : 0]), "n"(N))
template <int NI, int NH, int N>
__device__ __forceinline__ float gate_row(const float x0, const float x1, const float* wi, const float wb, const float h, const float* wh)
{
  float acc, tmp;
  if constexpr (NI == 1 && NH == 1)
    NAM_LROW_ASM(NAM_LROW_X0 NAM_LROW_H0);
  else if constexpr (NI == 1 && NH == 2)
    NAM_LROW_ASM(NAM_LROW_X0 NAM_LROW_H0 NAM_LROW_H1);
  else if constexpr (NI == 1 && NH == 3)
    NAM_LROW_ASM(NAM_LROW_X0 NAM_LROW_H0 NAM_LROW_H1 NAM_LROW_H2);
  else if constexpr (NI == 1)
    NAM_LROW_ASM(NAM_LROW_X0 NAM_LROW_H0 NAM_LROW_H1 NAM_LROW_H2 NAM_LROW_H3);
  else if constexpr (NH == 1)
    NAM_LROW_ASM(NAM_LROW_X0 NAM_LROW_X1 NAM_LROW_H0);
  else if constexpr (NH == 2)
    NAM_LROW_ASM(NAM_LROW_X0 NAM_LROW_X1 NAM_LROW_H0 NAM_LROW_H1);
  else if constexpr (NH == 3)
    NAM_LROW_ASM(NAM_LROW_X0 NAM_LROW_X1 NAM_LROW_H0 NAM_LROW_H1 NAM_LROW_H2);
  else
    NAM_LROW_ASM(NAM_LROW_X0 NAM_LROW_X1 NAM_LROW_H0 NAM_LROW_H1 NAM_LROW_H2 NAM_LROW_H3);
  return acc;
}
#undef NAM_LROW_ASM
#undef NAM_LROW_H3
#undef NAM_LROW_H2
#undef NAM_LROW_H1
#undef NAM_LROW_H0
#undef NAM_LROW_X1
#undef NAM_LROW_X0
#undef NAM_LROW_DPP
#undef NAM_LROW_TERM
} // namespace lrow

template <int NL, int NI, int NH, bool FAST>
__global__ __launch_bounds__(64) void nam_lstm_row_kernel(const float* __restrict__ blob, const LSTMArgs a)
{
  // The 64 steps of a block are unrolled (compile-time t): the input sample of step t is a DPP row broadcast out of
  // the lane / register that loaded it (lane q of the row keeps frames q, q + 16, q + 32, q + 48), so neither an LDS
  // read nor any address arithmetic sits in the step; the compiler folds the broadcasts (input, h of every unit) into
  // the FMAs that consume them. The head is NOT computed in the step: the top layer's h goes to LDS (one store per
  // step, immediate offset) and the 64 outputs are formed afterwards with lane = frame.
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  const int row = lane >> 4, q = lane & 15;
  const int u = q >> 2, k = q & 3; // hidden unit, gate (i, f, g, o)
  const int s0 = blockIdx.x * 4;
  const bool live = s0 + row < a.n_streams;
  const int stream = live ? (a.stream_map ? a.stream_map[s0 + row] : s0 + row) : 0;
  constexpr int H = NH; // hidden units (compile time: no broadcast / FMA is spent on padding units)
  const int out_ch = a.out_ch;
  const bool unit = u < H;
  float* const hs = lds; // [4 rows][NH][65]: the top layer's h of every step; then 64 + 64 floats nobody reads
  // sigmoid(x) = 0.5 tanh(x / 2) + 0.5 for gates i, f, o; tanh for g: act(x) = A T(B x) + C. B (1 or 0.5: exact) is
  // folded into the lane's weights.
  const float cA = k == 2 ? 1.0f : 0.5f, cB = cA, cC = k == 2 ? 0.0f : 0.5f;
  // the rational's numerator coefficients times A (gates) / times 1 (tanh of the cell state): lrow::act_affine
  constexpr float kRa = 2.45550750702956f, kRb = 0.893229853513558f, kRc = 0.821226666969744f, kRd = 2.44506634652299f;
  constexpr float kRa1 = (float)(2.45550750702956 - 2.44506634652299 * 0.821226666969744), kRa0 = (float)(2.45550750702956 - 2.44506634652299 * 0.893229853513558);
  const lrow::f2 knA = {kRc * cA, kRa1 * cA}, kcA = {kRb * cA, kRa0 * cA};
  const lrow::f2 kn1 = {kRc, kRa1}, kc1 = {kRb, kRa0};
  (void)kRa, (void)kRd;

  // this lane's gate row of every layer: bias, input weights, recurrent weights (zero rows for the padding unit)
  constexpr int NW = NI > NH ? NI : NH;
  float wb[NL], wi[NL][NW], wh[NL][NH];
#pragma unroll
  for (int l = 0; l < NL; l++)
  {
    const int I = l == 0 ? NI : H;
    const float* __restrict__ W = blob + a.layer_w[l] + (size_t)(k * H + (unit ? u : 0)) * (I + H);
    wb[l] = unit ? cB * blob[a.layer_b[l] + k * H + u] : 0.0f;
#pragma unroll
    for (int e = 0; e < NW; e++)
      wi[l][e] = (unit && e < I) ? cB * W[e] : 0.0f;
#pragma unroll
    for (int j = 0; j < NH; j++)
      wh[l][j] = unit ? cB * W[I + j] : 0.0f;
  }

  // persistent session (persist_wave.h): blocks come from commands, not from a frame count
  const bool pers = a.ps.ring != nullptr;
  PersistWave pw;
  unsigned cmd_off = 0;
  if (pers && !pw.begin(a.ps, (int)blockIdx.x, cmd_off))
  {
    pw.leave(a.ps, (int)blockIdx.x); // nothing to do
    return;
  }

  float* st = a.state + (size_t)stream * a.state_stride;
  float h[NL], c[NL];
#pragma unroll
  for (int l = 0; l < NL; l++)
  {
    h[l] = (live && unit) ? st[(l * 2 + 0) * H + u] : 0.0f;
    c[l] = (live && unit) ? st[(l * 2 + 1) * H + u] : 0.0f;
  }
  // where this lane's copy of the top h goes each step: gate lane 0 of a real unit -> its history row, else a dump slot
  const unsigned hs_slot = (k == 0 && unit) ? (unsigned)((row * NH + u) * 65) : (unsigned)(4 * NH * 65 + lane);

  for (int f0 = pers ? (int)cmd_off : 0;;)
  {
    const int nvalid = pers ? kBlock : min(kBlock, a.n_frames - f0);
    if (pers)
      pw.look_ahead(a.ps);
    // lane q of the row: frames q, q + 16, q + 32, q + 48 of the row's stream (16 lanes = 64 contiguous bytes)
    float xr[NI][4];
#pragma unroll
    for (int e = 0; e < NI; e++)
#pragma unroll
      for (int j = 0; j < 4; j++)
      {
        const int t = q + 16 * j;
        xr[e][j] = 0.0f;
        if (a.in && live && t < nvalid)
        {
          const float* px = a.in + ((size_t)stream * a.in_ch + e) * a.io_stride + f0 + t;
          xr[e][j] = pers ? persist_in(px) : *px;
        }
      }
    auto step = [&](auto t_tag, auto whole_tag) {
      constexpr int T = decltype(t_tag)::value;
      if constexpr (!decltype(whole_tag)::value)
        if (T >= nvalid) // (wavefront-uniform; a whole block — every block but a launch's ragged last one — runs the steps without the test)
          return;
#pragma unroll
      for (int l = 0; l < NL; l++)
      {
        // gate pre-activation of this lane's row (times B): b + Wi . in + Wh . h(t - 1)
        float pre = wb[l];
        if (l == 0)
          pre = lrow::gate_row<NI, NH, T % 16>(xr[0][T / 16], xr[NI - 1][T / 16], wi[0], wb[0], h[0], wh[0]);
        else
        {
          const float hb_ = h[l > 0 ? l - 1 : 0]; // the layer below's h(t)
          pre = fmaf(wi[l][0], lrow::row_bcast<0>(hb_), pre);
          if constexpr (NH > 1)
            pre = fmaf(wi[l][1], lrow::row_bcast<4>(hb_), pre);
          if constexpr (NH > 2)
            pre = fmaf(wi[l][2], lrow::row_bcast<8>(hb_), pre);
          if constexpr (NH > 3)
            pre = fmaf(wi[l][3], lrow::row_bcast<12>(hb_), pre);
        }
        // the recurrent terms, one chain (longer than an FMA's latency apart anyway), each ONE instruction. h(t - 1) was written
        // by the step before: two wait states in front of the first DPP read of it
        if (l > 0)
          lrow::recurrent_terms<NH>(pre, h[l], wh[l]);
        float late;
        float g = lrow::ratio<FAST>(pre, cA, knA, kcA, late);
        g = FAST ? fmaf(g, pre, cC) : g + cC;
        // the unit's gates meet in lanes 0 and 2 of its quad — i * g on one DPP multiply (quad_perm [2, 3, 0, 1]: lane 0 reads g,
        // lane 2 reads i), + f * c on one DPP multiply-add. Lanes 1 and 3 compute something nobody reads: h goes on from
        // lane 0 of the quad (row_newbcast 0 / 4 / 8 / 12), the state and the history row are written by it
        float cn;
        asm("s_nop 1\n\tv_mul_f32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1"
            : "=&v"(cn)
            : "v"(g), "v"(c[l]));
        // h = o * tanh_like(c): o * c on a DPP multiply of its own — placed behind the ratio's denominator (a fake operand):
        // well over two wait states behind c's write
        float r = lrow::ratio<FAST>(cn, 1.0f, kn1, kc1, late);
        float oc;
        asm("v_mul_f32_dpp %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(oc) : "v"(g), "v"(FAST ? cn : r), "v"(late));
        const float hn = FAST ? r * oc : oc;
        c[l] = cn;
        h[l] = hn;
      }
      hs[hs_slot + T] = h[NL - 1];
    };
    if (nvalid == kBlock)
      il::for_each_index([&](auto t_tag) { step(t_tag, std::true_type{}); }, std::make_integer_sequence<int, kBlock>{});
    else
      il::for_each_index([&](auto t_tag) { step(t_tag, std::false_type{}); }, std::make_integer_sequence<int, kBlock>{});
    // head: y[ch][t] = bh + Wh . h_top(t), lane = frame (coalesced stores)
    if (a.out)
      for (int r = 0; r < 4; r++)
      {
        const int s = __shfl(stream, r * 16);
        if (s0 + r >= a.n_streams)
          break;
        float hv[NH];
#pragma unroll
        for (int j = 0; j < NH; j++)
          hv[j] = hs[(r * NH + j) * 65 + lane];
        for (int ch = 0; ch < out_ch; ch++)
        {
          float y = blob[a.head_b + ch];
#pragma unroll
          for (int j = 0; j < NH; j++)
            y = fmaf(blob[a.head_w + ch * H + j], hv[j], y);
          if (lane < nvalid)
            a.out[((size_t)s * out_ch + ch) * a.io_stride + f0 + lane] = y;
        }
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
  if (live && unit && k == 0)
#pragma unroll
    for (int l = 0; l < NL; l++)
    {
      st[(l * 2 + 0) * H + u] = h[l];
      st[(l * 2 + 1) * H + u] = c[l];
    }
  if (pers)
    pw.leave(a.ps, (int)blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// Wide LSTM cells (5 .. 32 hidden units, 1 - 2 layers: the sizes real NAM LSTM captures use), ONE WAVEFRONT PER STREAM,
// TWO GATE ROWS PER LANE: lane 2u holds rows (i, f) of hidden unit u, lane 2u + 1 rows (g, o) — every weight of the
// model lives in registers as pairs, so a time step is one v_pk_fma_f32 per input and lane (both rows at once, the input
// as a scalar operand). A layer's new h is broadcast ONCE per step with v_readlane into scalar registers: it feeds the
// layer above in this step and the layer's own recurrence in the next. The pair of lanes of a unit swaps its two
// activations through one quad_perm DPP move each; c and h are kept in both. sigmoid(x) = 0.5 tanh(x / 2) + 0.5 as in the
// gate-row kernel (the 0.5 folded into the weights: exact). The top layer's h goes to LDS every step and the block's 64
// outputs are formed afterwards with lane = frame. 1,024 streams = 1,024 wavefronts (the 16-streams-per-wavefront MFMA
// kernel keeps 64 busy and issues ~3 k cycles of matrix work per step for a 2 x 18 model).
// Reference: NAM/lstm.cpp:31-68 (cell), :103-168 (process), gate order i, f, g, o; fast forms :48-58. State layout
// [layer][h | c][H] shared with the other LSTM kernels.
// ------------------------------------------------------------------------------------------------
namespace lwide
{
using f2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ float pair_swap(float v) // the other lane of the (2u, 2u + 1) pair
{
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_bcast(float v, int src) // lane `src` (compile-time or scalar) -> a scalar register
{
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}
// tanh (or the reference's fast_tanh rational, activations.h:29-41) of both rows at once: packed arithmetic, two rcp
template <bool FAST>
__device__ __forceinline__ f2 tanh2(const f2 x)
{
  if constexpr (FAST)
  {
    // the rearranged rational of the gate-row kernel (lrow::ratio): q = x^2 + d, |x + e x |x|| = |x| (1 + e |x|), numerator / x =
    // t1 q + (t2 - d t1) — eight packed instructions, two rcp, two abs
    constexpr float a1 = (float)(2.45550750702956 - 2.44506634652299 * 0.821226666969744), a0 = (float)(2.45550750702956 - 2.44506634652299 * 0.893229853513558);
    const f2 ax = __builtin_elementwise_abs(x);
    const f2 kd = f2{2.44506634652299f, 2.44506634652299f};
    const f2 q = __builtin_elementwise_fma(ax, ax, kd);
    const f2 w = __builtin_elementwise_fma(f2{0.814642734961073f, 0.814642734961073f}, ax, f2{1.0f, 1.0f});
    const f2 t1 = __builtin_elementwise_fma(f2{0.821226666969744f, 0.821226666969744f}, ax, f2{0.893229853513558f, 0.893229853513558f});
    const f2 t2 = __builtin_elementwise_fma(f2{a1, a1}, ax, f2{a0, a0});
    const f2 den = __builtin_elementwise_fma(q, ax * w, kd);
    const f2 n = __builtin_elementwise_fma(t1, q, t2);
    return (n * f2{mf::rcp(den[0]), mf::rcp(den[1])}) * x;
  }
  else
  {
    const f2 y = x * f2{2.885390081777927f, 2.885390081777927f}; // exp(2x) = 2^(2x log2 e)
    const f2 e = f2{__builtin_amdgcn_exp2f(y[0]), __builtin_amdgcn_exp2f(y[1])} + f2{1.0f, 1.0f};
    return __builtin_elementwise_fma(f2{-2.0f, -2.0f}, f2{mf::rcp(e[0]), mf::rcp(e[1])}, f2{1.0f, 1.0f});
  }
}
} // namespace lwide

template <int NL, int NI, int NH, bool FAST>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) void nam_lstm_wide_kernel(
  const float* __restrict__ blob, const LSTMArgs a)
{
  using lwide::f2;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int HP = NH + 4; // history row pitch: conflict-free b128 reads with lane = frame
  const int lane = threadIdx.x;
  const int u = lane >> 1;
  const bool odd = (lane & 1) != 0;
  const int H = a.hidden; // <= NH (NH: the next multiple of 4; padding units have zero rows and stay at h = c = 0)
  const bool unit = u < H;
  const int stream = a.stream_map ? a.stream_map[blockIdx.x] : (int)blockIdx.x;
  const int out_ch = a.out_ch;
  float* const hist = lds; // [64 steps][HP]: the top layer's h
  float* const hw = lds + kBlock * HP; // head weights [out_ch][NH] (zero padded), then the head bias [out_ch]

  // this lane's two gate rows of every layer (row 0: i or g, row 1: f or o), as pairs: input weights, recurrent
  // weights, bias. Rows that feed a sigmoid carry the 0.5 of sigmoid(x) = 0.5 tanh(x / 2) + 0.5.
  const int g0 = odd ? 2 : 0, g1 = odd ? 3 : 1;
  const float s0 = odd ? 1.0f : 0.5f; // row 0: g -> tanh, i -> sigmoid; row 1 (f, o) is always a sigmoid
  constexpr int NIN = NI > NH ? NI : NH;
  f2 wi[NL][NIN], wh[NL][NH], wb[NL];
#pragma unroll
  for (int l = 0; l < NL; l++)
  {
    const int I = l == 0 ? NI : H;
    const float* __restrict__ W0 = blob + a.layer_w[l] + (size_t)(g0 * H + (unit ? u : 0)) * (I + H);
    const float* __restrict__ W1 = blob + a.layer_w[l] + (size_t)(g1 * H + (unit ? u : 0)) * (I + H);
    wb[l] = unit ? f2{s0 * blob[a.layer_b[l] + g0 * H + u], 0.5f * blob[a.layer_b[l] + g1 * H + u]} : f2{0.0f, 0.0f};
#pragma unroll
    for (int e = 0; e < NIN; e++)
      wi[l][e] = (unit && e < I) ? f2{s0 * W0[e], 0.5f * W1[e]} : f2{0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < NH; j++)
      wh[l][j] = (unit && j < H) ? f2{s0 * W0[I + j], 0.5f * W1[I + j]} : f2{0.0f, 0.0f};
  }
  const float a0 = odd ? 1.0f : 0.5f, b0 = odd ? 0.0f : 0.5f; // row 0's activation = a0 T(z) + b0
  for (int i = lane; i < out_ch * NH; i += kBlock)
    hw[i] = (i % NH) < H ? blob[a.head_w + (i / NH) * H + (i % NH)] : 0.0f;
  if (lane < out_ch)
    hw[out_ch * NH + lane] = blob[a.head_b + lane];

  // persistent session (persist_wave.h): blocks come from commands, not from a frame count
  const bool pers = a.ps.ring != nullptr;
  PersistWave pw;
  unsigned cmd_off = 0;
  if (pers && !pw.begin(a.ps, (int)blockIdx.x, cmd_off))
  {
    pw.leave(a.ps, (int)blockIdx.x); // nothing to do
    return;
  }

  float* const st = a.state + (size_t)stream * a.state_stride;
  float h[NL], c[NL];
  float hb[NL][NH]; // every unit's h of every layer, wavefront-uniform (scalar registers)
#pragma unroll
  for (int l = 0; l < NL; l++)
  {
    h[l] = unit ? st[(l * 2 + 0) * H + u] : 0.0f;
    c[l] = unit ? st[(l * 2 + 1) * H + u] : 0.0f;
#pragma unroll
    for (int j = 0; j < NH; j++)
      hb[l][j] = lwide::lane_bcast(h[l], 2 * j);
  }
  const float* const in = a.in ? a.in + (size_t)stream * a.in_ch * a.io_stride : nullptr;
  float* const out = a.out ? a.out + (size_t)stream * out_ch * a.io_stride : nullptr;

  for (int f0 = pers ? (int)cmd_off : 0;;)
  {
    const int nvalid = pers ? kBlock : min(kBlock, a.n_frames - f0);
    if (pers)
      pw.look_ahead(a.ps);
    float xv[NI]; // lane = frame
#pragma unroll
    for (int e = 0; e < NI; e++)
    {
      xv[e] = 0.0f;
      if (in && lane < nvalid)
        xv[e] = pers ? persist_in(in + (size_t)e * a.io_stride + f0 + lane) : in[(size_t)e * a.io_stride + f0 + lane];
    }
#pragma unroll 1 // (64 unrolled steps of ~200 instructions would not fit the instruction cache)
    for (int t = 0; t < nvalid; t++)
    {
#pragma unroll
      for (int l = 0; l < NL; l++)
      {
        // both rows' pre-activations (times their 1 or 0.5): W_in . in + W_h . h(t - 1) + b. Two partial sums: a
        // v_pk_fma_f32 that depends on the one right before it costs a wait state (and an s_nop) each time
        f2 z = f2{0.0f, 0.0f}, zb = f2{0.0f, 0.0f};
        if (l == 0)
        {
#pragma unroll
          for (int e = 0; e < NI; e++)
          {
            const float x = lwide::lane_bcast(xv[e], t);
            z = __builtin_elementwise_fma(wi[0][e], f2{x, x}, z);
          }
        }
        else
        {
#pragma unroll
          for (int j = 0; j < NH; j += 2)
          {
            z = __builtin_elementwise_fma(wi[l][j], f2{hb[l - 1][j], hb[l - 1][j]}, z);
            zb = __builtin_elementwise_fma(wi[l][j + 1], f2{hb[l - 1][j + 1], hb[l - 1][j + 1]}, zb);
          }
        }
#pragma unroll
        for (int j = 0; j < NH; j += 2)
        {
          zb = __builtin_elementwise_fma(wh[l][j], f2{hb[l][j], hb[l][j]}, zb);
          z = __builtin_elementwise_fma(wh[l][j + 1], f2{hb[l][j + 1], hb[l][j + 1]}, z);
        }
        z += zb;
        z += wb[l];
        const f2 r = __builtin_elementwise_fma(f2{a0, 0.5f}, lwide::tanh2<FAST>(z), f2{b0, 0.5f});
        const float r0 = r[0], r1 = r[1];
        const float p0 = lwide::pair_swap(r0), p1 = lwide::pair_swap(r1);
        // even lane: (i, f) own, (g, o) from the partner; odd lane the other way round
        const float gi = odd ? p0 : r0, gf = odd ? p1 : r1, gg = odd ? r0 : p0, go = odd ? r1 : p1;
        const float cn = fmaf(gf, c[l], gi * gg);
        const float hn = go * lrow::tanh_like<FAST>(cn);
        c[l] = cn;
        h[l] = hn;
#pragma unroll
        for (int j = 0; j < NH; j++)
          hb[l][j] = lwide::lane_bcast(hn, 2 * j);
      }
      if (!odd && u < NH)
        hist[t * HP + u] = h[NL - 1];
    }
    // head: y[ch][t] = Wh . h_top(t) + bh, lane = frame (coalesced stores)
    if (out)
    {
      float hv[NH];
#pragma unroll
      for (int q = 0; q < NH / 4; q++)
      {
        const mf::f4 v = *reinterpret_cast<const mf::f4*>(hist + lane * HP + 4 * q);
        hv[4 * q] = v[0], hv[4 * q + 1] = v[1], hv[4 * q + 2] = v[2], hv[4 * q + 3] = v[3];
      }
      for (int ch = 0; ch < out_ch; ch++)
      {
        float y = 0.0f;
#pragma unroll
        for (int j = 0; j < NH; j++)
          y = fmaf(hw[ch * NH + j], hv[j], y);
        y += hw[out_ch * NH + ch];
        if (lane < nvalid)
          out[(size_t)ch * a.io_stride + f0 + lane] = y;
      }
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
  if (unit && !odd)
#pragma unroll
    for (int l = 0; l < NL; l++)
    {
      st[(l * 2 + 0) * H + u] = h[l];
      st[(l * 2 + 1) * H + u] = c[l];
    }
  if (pers)
    pw.leave(a.ps, (int)blockIdx.x);
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

int lstm_lds_bytes(const LSTMArgs& a)
{
  const long floats = (long)(a.in_ch + a.out_ch) * kBlock * 65 + 2l * a.n_layers * a.hidden * kBlock + 4l * a.hidden * kBlock;
  return (int)std::min<long>(floats * (long)sizeof(float), 1l << 30);
}
// floats of global scratch nam_lstm_kernel<true> needs for `n_streams` (0: the cell fits LDS)
long lstm_scratch_floats(const LSTMArgs& a)
{
  if (lstm_lds_bytes(a) <= 160 * 1024)
    return 0;
  const long n_blocks = (a.n_streams + kBlock - 1) / kBlock;
  return n_blocks * (long)(2 * a.n_layers + 4) * a.hidden * kBlock;
}

hipError_t launch_lstm(const LSTMArgs& a, hipStream_t stream)
{
  const int n_blocks = (a.n_streams + kBlock - 1) / kBlock;
  if (lstm_scratch_floats(a) > 0)
  {
    if (!a.scratch)
      return hipErrorInvalidValue;
    const int io_bytes = (a.in_ch + a.out_ch) * kBlock * 65 * (int)sizeof(float);
    if (io_bytes > 160 * 1024)
      return hipErrorInvalidValue;
    static DynamicLdsLimit lim;
    if (io_bytes > 64 * 1024)
    {
      const hipError_t e = lim.ensure(reinterpret_cast<const void*>(nam_lstm_kernel<true>), 160 * 1024);
      if (e != hipSuccess)
        return e;
    }
    hipLaunchKernelGGL(nam_lstm_kernel<true>, dim3(n_blocks), dim3(64), io_bytes, stream, a.blob, a);
    return hipGetLastError();
  }
  const int lds_bytes = lstm_lds_bytes(a);
  if (lds_bytes > 64 * 1024)
  {
    static DynamicLdsLimit lim;
    const hipError_t e = lim.ensure(reinterpret_cast<const void*>(nam_lstm_kernel<false>), 160 * 1024);
    if (e != hipSuccess)
      return e;
  }
  hipLaunchKernelGGL(nam_lstm_kernel<false>, dim3(n_blocks), dim3(64), lds_bytes, stream, a.blob, a);
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
  if (a.fast) \
    hipLaunchKernelGGL((nam_lstm_mfma_reg_kernel<NL, NT, true>), dim3(n_blocks), dim3(64), io_bytes, stream, a.blob, a); \
  else \
    hipLaunchKernelGGL((nam_lstm_mfma_reg_kernel<NL, NT, false>), dim3(n_blocks), dim3(64), io_bytes, stream, a.blob, a)
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

bool lstm_row_eligible(const LSTMArgs& a)
{
  return a.hidden >= 1 && a.hidden <= 4 && a.n_layers >= 1 && a.n_layers <= 2 && a.input_size >= 1 && a.input_size <= 2
         && a.in_ch == a.input_size && a.out_ch >= 1 && a.out_ch <= 16;
}

hipError_t launch_lstm_row(const LSTMArgs& a, hipStream_t stream)
{
  if (!lstm_row_eligible(a))
    return hipErrorInvalidValue;
  const int n_blocks = (a.n_streams + 3) / 4;
  const int lds_bytes = (4 * a.hidden * 65 + 128) * (int)sizeof(float); // h history of the block + the dump slots
#define NAM_LSTM_ROW_H(NL, NI, NH) \
  if (a.fast) \
    nam_launch((nam_lstm_row_kernel<NL, NI, NH, true>), dim3(n_blocks), dim3(64), lds_bytes, stream, a.blob, a); \
  else \
    nam_launch((nam_lstm_row_kernel<NL, NI, NH, false>), dim3(n_blocks), dim3(64), lds_bytes, stream, a.blob, a)
#define NAM_LSTM_ROW(NL, NI) \
  switch (a.hidden) \
  { \
    case 1: NAM_LSTM_ROW_H(NL, NI, 1); break; \
    case 2: NAM_LSTM_ROW_H(NL, NI, 2); break; \
    case 3: NAM_LSTM_ROW_H(NL, NI, 3); break; \
    default: NAM_LSTM_ROW_H(NL, NI, 4); break; \
  }
  switch (a.n_layers * 10 + a.input_size)
  {
    case 11: NAM_LSTM_ROW(1, 1); break;
    case 12: NAM_LSTM_ROW(1, 2); break;
    case 21: NAM_LSTM_ROW(2, 1); break;
    default: NAM_LSTM_ROW(2, 2); break;
  }
#undef NAM_LSTM_ROW
#undef NAM_LSTM_ROW_H
  return hipGetLastError();
}

bool lstm_wide_eligible(const LSTMArgs& a)
{
  return a.hidden >= 5 && a.hidden <= 32 && a.n_layers >= 1 && a.n_layers <= 2 && a.input_size >= 1 && a.input_size <= 2
         && a.in_ch == a.input_size && a.out_ch >= 1 && a.out_ch <= 16;
}

hipError_t launch_lstm_wide(const LSTMArgs& a, hipStream_t stream)
{
  if (!lstm_wide_eligible(a))
    return hipErrorInvalidValue;
  const int nh = (a.hidden + 3) & ~3;
  const int lds_bytes = (kBlock * (nh + 4) + a.out_ch * nh + a.out_ch) * (int)sizeof(float);
#define NAM_LSTM_WIDE_H(NL, NI, NH) \
  if (a.fast) \
    nam_launch((nam_lstm_wide_kernel<NL, NI, NH, true>), dim3(a.n_streams), dim3(64), lds_bytes, stream, a.blob, a); \
  else \
    nam_launch((nam_lstm_wide_kernel<NL, NI, NH, false>), dim3(a.n_streams), dim3(64), lds_bytes, stream, a.blob, a)
#define NAM_LSTM_WIDE(NL, NI) \
  switch (nh) \
  { \
    case 8: NAM_LSTM_WIDE_H(NL, NI, 8); break; \
    case 12: NAM_LSTM_WIDE_H(NL, NI, 12); break; \
    case 16: NAM_LSTM_WIDE_H(NL, NI, 16); break; \
    case 20: NAM_LSTM_WIDE_H(NL, NI, 20); break; \
    case 24: NAM_LSTM_WIDE_H(NL, NI, 24); break; \
    case 28: NAM_LSTM_WIDE_H(NL, NI, 28); break; \
    default: NAM_LSTM_WIDE_H(NL, NI, 32); break; \
  }
  switch (a.n_layers * 10 + a.input_size)
  {
    case 11: NAM_LSTM_WIDE(1, 1); break;
    case 12: NAM_LSTM_WIDE(1, 2); break;
    case 21: NAM_LSTM_WIDE(2, 1); break;
    default: NAM_LSTM_WIDE(2, 2); break;
  }
#undef NAM_LSTM_WIDE
#undef NAM_LSTM_WIDE_H
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
