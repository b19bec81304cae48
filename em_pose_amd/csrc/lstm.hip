// LSTM time steps on the fp32 matrix cores (reference nn/layers.py:133-157: nn.LSTM, gate order i,f,g,o, stacked
// layers, pack_padded_sequence semantics).
//
// One launch advances the whole layer stack by one "wavefront" step s: layer l processes time step t = s - l, so
// all layers run concurrently (layer l at step t only needs layer l-1 at step t, produced by the previous launch,
// and its own state after step t-1).  For every (layer, step) the gate pre-activations are
//     gates = in_t . W_ih^T + h_{t-1} . W_hh^T + (b_ih + b_hh)
// i.e. one GEMM whose K dimension is the concatenation of two operand "segments"; the cell non-linearities are the
// epilogue.  Nothing but h, c and the last layer's output ever touches memory (no gate buffer, no separate input
// projection).
//
// Every blockIdx.z is one "unit" = one (layer, direction). Units either read a stored input sequence (layer 0, and the
// layers of a bidirectional stack, whose input is the concatenated output sequence of the previous layer) or the hidden
// state another unit produced in the previous launch (uni-directional stacks: the wavefront above).  A reverse unit
// visits, at its step k, time len_b-1-k of every row b (packed-sequence semantics for ragged batches).
//
// Tiling: v_mfma_f32_16x16x4_f32.  A wave owns 32 batch rows x 16 hidden units and keeps 2 x 4 accumulators
// (row half x gate), so i/f/g/o of one (row, unit) sit in the same lane.  A block is 2x2 waves = 64 rows x 32 units
// (128 weight rows per K tile) -> for B=1024, H=512: 256 blocks per layer, one wave per SIMD and layer.
// K tiles (32 wide) are register-prefetched one tile ahead; since a dot product does not care about the order of k,
// each 16-lane group takes 4 consecutive k of a 16-wide chunk so that one ds_read_b128 feeds four MFMAs.
#include "kernels.h"

namespace empose {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int LBK = 32;
constexpr int LLD = LBK + 4;
constexpr int LROWS = 64;   // batch rows per block
constexpr int LUNITS = 32;  // hidden units per block

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

struct Seg {
  const float* a; int lda;   // [B][lda]
  const float* w; int ldw;   // [4H][ldw]
  int K;
  // per-row time offset (reverse units): row b reads a + b*lda + trow(b)*tstride; tstride == 0 => none
  const int* lens; int k; int tstride;
};

__device__ __forceinline__ void lstm_load(const Seg& sg, int k0, int m0, int j0, int B, int H, int tid,
                                          float4 (&ra)[2], float4 (&rb)[4]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int slot = tid + i * 256;
    const int r = slot >> 3, c4 = (slot & 7) * 4;
    ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m0 + r < B && k0 + c4 < sg.K) {
      size_t off = (size_t)(m0 + r) * sg.lda + k0 + c4;
      if (sg.tstride) {
        const int tr = sg.lens[m0 + r] - 1 - sg.k;   // reverse unit: time visited by this row
        off += (size_t)(tr > 0 ? tr : 0) * sg.tstride;
      }
      ra[i] = *reinterpret_cast<const float4*>(sg.a + off);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int slot = tid + i * 256;
    const int r = slot >> 3, c4 = (slot & 7) * 4;  // r in [0,128): gate = r >> 5, unit = r & 31
    const int unit = j0 + (r & 31);
    rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (unit < H && k0 + c4 < sg.K)
      rb[i] = *reinterpret_cast<const float4*>(sg.w + (size_t)((r >> 5) * H + unit) * sg.ldw + k0 + c4);
  }
}

__device__ __forceinline__ void lstm_store(float* As, float* Bs, int tid, const float4 (&ra)[2], const float4 (&rb)[4]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int slot = tid + i * 256;
    *reinterpret_cast<float4*>(As + (slot >> 3) * LLD + (slot & 7) * 4) = ra[i];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int slot = tid + i * 256;
    *reinterpret_cast<float4*>(Bs + (slot >> 3) * LLD + (slot & 7) * 4) = rb[i];
  }
}

__global__ __launch_bounds__(256) void lstm_wave_kernel(LstmWaveArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[(LROWS + 4 * LUNITS) * LLD];
  float* As = lds;
  float* Bs = lds + LROWS * LLD;

  const LstmUnitArgs& L = a.unit[blockIdx.z];
  const int t = a.s - L.t_offset;     // step index k of this unit (time index for forward units)
  if (t < 0 || t >= a.F) return;
  const int H = a.H, B = a.B;
  // blockIdx.x (the fast index, which also selects the XCD: workgroup b runs on XCD b % 8) walks the UNIT tiles, so
  // one XCD's L2 holds the W_ih/W_hh rows of two unit tiles and streams the (smaller) h/x rows of all batch tiles.
  const int j0 = blockIdx.x * LUNITS, m0 = blockIdx.y * LROWS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wrow = wave >> 1, wcol = wave & 1;
  const int l15 = lane & 15, lq = lane >> 4;
  const bool rev = L.reverse != 0;

  Seg segs[2];
  // segment 0: the unit's input at this step
  segs[0].lens = a.seq_lengths; segs[0].k = t; segs[0].tstride = 0;
  if (L.in_from < 0) {
    segs[0].lda = a.F * L.in_ld;
    if (rev) { segs[0].a = L.in_seq; segs[0].tstride = L.in_ld; }
    else segs[0].a = L.in_seq + (size_t)t * L.in_ld;
  } else {
    segs[0].a = a.unit[L.in_from].h[(t + 1) & 1]; segs[0].lda = H;
  }
  segs[0].w = L.w_ih; segs[0].ldw = L.in_k; segs[0].K = L.in_k;
  // segment 1: own hidden state after the previous step
  segs[1].a = L.h[t & 1]; segs[1].lda = H; segs[1].w = L.w_hh; segs[1].ldw = H; segs[1].K = H;
  segs[1].lens = nullptr; segs[1].k = 0; segs[1].tstride = 0;

  f32x4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk0 = (segs[0].K + LBK - 1) / LBK, nk1 = (segs[1].K + LBK - 1) / LBK;
  const int nk = nk0 + nk1;
  float4 ra[2], rb[4];
  lstm_load(segs[0], 0, m0, j0, B, H, tid, ra, rb);
  lstm_store(As, Bs, tid, ra, rb);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) {
      const int nx = kt + 1;
      if (nx < nk0) lstm_load(segs[0], nx * LBK, m0, j0, B, H, tid, ra, rb);
      else lstm_load(segs[1], (nx - nk0) * LBK, m0, j0, B, H, tid, ra, rb);
    }
#pragma unroll
    for (int kk = 0; kk < LBK / 16; ++kk) {
      float4 av[2], bv[4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        av[i] = *reinterpret_cast<const float4*>(As + (wrow * 32 + i * 16 + l15) * LLD + kk * 16 + lq * 4);
#pragma unroll
      for (int g = 0; g < 4; ++g)
        bv[g] = *reinterpret_cast<const float4*>(Bs + (g * LUNITS + wcol * 16 + l15) * LLD + kk * 16 + lq * 4);
      // k-element outermost: consecutive MFMAs go to 8 different accumulators, so the 40-cycle dependent latency of
      // v_mfma_f32_16x16x4_f32 (issue interval 32) never stalls the pipe.
#define LSTM_MFMA_STEP(E)                                                                                      \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int g = 0; g < 4; ++g)                  \
      acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].E, bv[g].E, acc[i][g], 0, 0, 0);
      LSTM_MFMA_STEP(x)
      LSTM_MFMA_STEP(y)
      LSTM_MFMA_STEP(z)
      LSTM_MFMA_STEP(w)
#undef LSTM_MFMA_STEP
    }
    __syncthreads();
    if (kt + 1 < nk) {
      lstm_store(As, Bs, tid, ra, rb);
      __syncthreads();
    }
  }

  // Epilogue. C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + r.
  // The unit descriptor lives in kernarg memory; everything the loop needs is copied into registers first, because the
  // stores below could alias it as far as the compiler knows (a scalar reload + wait per element otherwise).
  const int unit = j0 + wcol * 16 + l15;
  if (unit >= H) return;
  const float* __restrict__ bias = L.bias;
  const float bi = bias[unit], bf = bias[H + unit], bg = bias[2 * H + unit], bo = bias[3 * H + unit];
  const float* __restrict__ h_prev = L.h[t & 1];
  float* __restrict__ h_next = L.h[(t + 1) & 1];
  float* __restrict__ cst = L.c;
  float* __restrict__ yout = L.y;
  const int* __restrict__ lens = a.seq_lengths;
  const int F = a.F;
  const long y_ld = L.y_ld, y_col = L.y_col;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + wrow * 32 + i * 16 + lq * 4 + r;
      if (row >= B) continue;
      const size_t hc = (size_t)row * H + unit;
      const int len = lens ? lens[row] : F;
      const bool live = t < len;
      const int t_out = (rev && live) ? len - 1 - t : t;   // a finished reverse row zero-fills the padded slot t
      float h_new;
      if (live) {
        const float c_new = sigmoidf_(acc[i][1][r] + bf) * cst[hc] + sigmoidf_(acc[i][0][r] + bi) * tanhf(acc[i][2][r] + bg);
        h_new = sigmoidf_(acc[i][3][r] + bo) * tanhf(c_new);
        cst[hc] = c_new;
        h_next[hc] = h_new;
      } else {
        h_next[hc] = h_prev[hc];
        h_new = 0.f;
      }
      if (yout) yout[((size_t)row * F + t_out) * y_ld + y_col + unit] = h_new;
    }
}

hipError_t launch_lstm_wave(const LstmWaveArgs& a, hipStream_t stream) {
  dim3 grid((a.H + LUNITS - 1) / LUNITS, (a.B + LROWS - 1) / LROWS, a.n_units);
  hipLaunchKernelGGL(lstm_wave_kernel, grid, dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace empose
