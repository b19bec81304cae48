// LSTM time steps on the fp32 matrix cores (reference nn/layers.py:133-157: nn.LSTM, gate order i,f,g,o, stacked
// layers, optionally bidirectional, pack_padded_sequence semantics).
//
// One launch advances the whole layer stack by one "wavefront" step s: layer l processes time step t = s - l, so
// all layers run concurrently (layer l at step t only needs layer l-1 at step t, produced by the previous launch,
// and its own state after step t-1).  For every (layer, step) the gate pre-activations are
//     gates = in_t . W_ih^T + h_{t-1} . W_hh^T + (b_ih + b_hh)
// i.e. one GEMM whose K dimension is the concatenation of two operand "segments"; the cell non-linearities are the
// epilogue.  Nothing but h, c and the last layer's output ever touches memory (no gate buffer, no separate input
// projection).
//
// A "unit" is one (layer, direction). Units either read a stored input sequence (layer 0, and the layers of a
// bidirectional stack, whose input is the concatenated output sequence of the previous layer) or the hidden state
// another unit produced in the previous launch (uni-directional stacks: the wavefront above).  A reverse unit visits, at
// its step k, time len_b-1-k of every row b (packed-sequence semantics for ragged batches).
//
// Tiling: v_mfma_f32_16x16x4_f32.  A wave owns 32 batch rows x 16 hidden units and keeps 2 x 4 accumulators
// (row half x gate), so i/f/g/o of one (row, unit) sit in the same lane; a block is 2x2 waves = 64 rows x 32 units.
// Since a dot product does not care about the order of k, each 16-lane group takes 4 consecutive k of a 16-wide chunk
// so that one ds_read_b128 feeds four MFMAs.
#include <type_traits>

#include "kernels.h"
#include "lane_reduce.h"

namespace empose {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------
// The wavefront kernel.
//
// A block owns one (64 batch rows) x (32 hidden units x 4 gates) tile and walks through a CHAIN of units one after the
// other -- for the stacked wavefront: layer 0 (K = input + H) then layer 1 (K = 2H) -- so every block does the same
// amount of work whatever the layers' K, and the 256 blocks of B = 1024, H = 512 are one block per CU.  The K tiles of
// the whole chain form one flat stream through a double-buffered LDS (2 x 48 KB, 64-wide K tiles, unpadded rows with
// the 16-byte chunks XOR-swizzled by row so that the ds_read_b128 lane groups of the 16x16x4 operand layout are
// conflict-free), one barrier per K tile, global loads two tiles ahead, and the instruction interleaving pinned with
// sched_group_barrier: with a single wave per SIMD nothing else hides a stall.
//   chunk 0 (k 0..15) : 32 MFMAs | fragment reads of chunk 1 | LDS writes of tile j+1 (fetched during tile j-1)
//   chunk 1           : 32 MFMAs | fragment reads of chunk 2 | global loads of tile j+2
//   chunk 2           : 32 MFMAs | fragment reads of chunk 3
//   barrier
//   chunk 3           : 32 MFMAs | fragment reads of chunk 0 of tile j+1
// The cell non-linearities run when the stream leaves a unit.
// ---------------------------------------------------------------------------------------------------------------
namespace lc {
constexpr int BM = 64, BU = 32, BK = 64;
constexpr int ROWS = BM + 4 * BU;
constexpr int STAGE = ROWS * BK;
constexpr size_t LDS_BYTES = 2 * (size_t)STAGE * sizeof(float);
constexpr int SG_MFMA = 0x008, SG_VMEM_RD = 0x020, SG_DS_RD = 0x100, SG_DS_WR = 0x200;
}  // namespace lc

#define LSTM_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

#ifdef EMPOSE_LSTM_TRACE   // dev lab only: per-phase shader-clock stamps of block (0,0,0)
__device__ long long g_lstm_trace[128];
#define LSTM_STAMP(i) \
  if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) g_lstm_trace[(i)] = clock64();
#else
#define LSTM_STAMP(i)
#endif

// Fast cell non-linearities: v_exp_f32 / v_rcp_f32 (about 1 ulp each); absolute error ~1e-7, far inside the 1e-4 parity
// budget, and the unit finish is no longer a visible fraction of the launch.
__device__ __forceinline__ float fsigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float ftanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }

typedef const __attribute__((address_space(1))) f32x4* gvec_t;
typedef const __attribute__((address_space(1))) char* gbyte_t;

__global__ __launch_bounds__(256) void lstm_chain_kernel(LstmWaveArgs a) {
  using namespace lc;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  LSTM_STAMP(0)
  const int H = a.H, B = a.B, F = a.F;
  const int j0 = blockIdx.x * BU, m0 = blockIdx.y * BM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wrow = wave >> 1, wcol = wave & 1;
  const int l15 = lane & 15, lq = lane >> 4;
  const int seg_beg = a.z_beg[blockIdx.z], n_seg = a.z_cnt[blockIdx.z];
  if (n_seg == 0) return;

  // ---- per-thread constants
  const int lr = tid >> 4, c16 = tid & 15;                 // global side: row lr + 16 i, 16-byte chunk c16 of the K tile
  int a_row[4], a_len[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = m0 + lr + 16 * i;
    a_row[i] = r < B ? r : B - 1;
    a_len[i] = a.seq_lengths ? a.seq_lengths[a_row[i]] : F;
  }
  int w_row[8];                                            // gate * H + unit of the thread's 8 weight rows
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int wr = lr + 16 * i, un = j0 + (wr & 31);
    w_row[i] = (wr >> 5) * H + (un < H ? un : H - 1);
  }
  const int wr_ofs = lr * BK + ((c16 ^ (lr & 15)) << 2);   // LDS write offset of piece 0 (floats); piece i: + 16 i rows
  const int a_rd = (wrow * 32 + l15) * BK;                 // fragment rows (floats); row tile i: + 16 rows
  const int b_rd = (BM + wcol * 16 + l15) * BK;            // gate g: + 32 rows
  int sw[4];                                               // swizzled chunk offset of k-chunk ch for this lane
#pragma unroll
  for (int ch = 0; ch < 4; ++ch) sw[ch] = ((((ch << 2) ^ (l15 & 12)) | (lq ^ (l15 & 3))) << 2);

  f32x4 g[12];
  bool g_ok = true;
  f32x4 fa[2][2], fb[2][4];
  f32x4 acc[2][4];

  // ---- the load side of the tile stream: current segment (scalar registers), the thread's byte offsets into its two
  // operands (recomputed when the stream enters a segment) and the k tile within the segment
  LstmSeg ld = a.seg[seg_beg];
  int ld_seg = 0, ld_kt = 0;
  unsigned a_off[4], w_off[8];
  auto seg_offsets = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int tr = a_len[i] - 1 - ld.k;
      a_off[i] = (unsigned)(((long)a_row[i] * ld.lda + (long)(tr > 0 ? tr : 0) * ld.tstride + c16 * 4) * 4);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) w_off[i] = (unsigned)(((long)w_row[i] * ld.ldw + c16 * 4) * 4);
  };
  seg_offsets();
  auto gload = [&]() -> bool {   // fetch the tile under the load cursor; returns whether it is ragged
    g_ok = ld_kt * BK + c16 * 4 < ld.K;
    const unsigned back = g_ok ? 0u : (unsigned)(c16 * 16);   // lanes past K re-read chunk 0 of the tile; zeroed later
    gbyte_t pa = (gbyte_t)(ld.a + ld_kt * BK);
    gbyte_t pw = (gbyte_t)(ld.w + ld_kt * BK);
#pragma unroll
    for (int i = 0; i < 4; ++i) g[i] = *(gvec_t)(pa + (a_off[i] - back));
#pragma unroll
    for (int i = 0; i < 8; ++i) g[4 + i] = *(gvec_t)(pw + (w_off[i] - back));
    return (ld_kt + 1) * BK > ld.K;
  };
  auto ld_advance = [&]() {      // uniform; past the end of the stream it parks on a valid tile that is never used
    if (++ld_kt == ld.ntiles) {
      ld_kt = 0;
      if (ld_seg + 1 < n_seg) { ++ld_seg; ld = a.seg[seg_beg + ld_seg]; seg_offsets(); }
    }
  };
  auto gzero = [&]() {
#pragma unroll
    for (int i = 0; i < 12; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) g[i][e] = g_ok ? g[i][e] : 0.f;
  };
  auto lwrite = [&](float* st) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(st + wr_ofs + i * 16 * BK) = g[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(st + BM * BK + wr_ofs + i * 16 * BK) = g[4 + i];
  };
  auto fread = [&](const float* st, int ch, f32x4 (&fa_)[2], f32x4 (&fb_)[4]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) fa_[i] = *reinterpret_cast<const f32x4*>(st + a_rd + i * 16 * BK + sw[ch]);
#pragma unroll
    for (int q = 0; q < 4; ++q) fb_[q] = *reinterpret_cast<const f32x4*>(st + b_rd + q * 32 * BK + sw[ch]);
  };
  auto mma = [&](const f32x4 (&fa_)[2], const f32x4 (&fb_)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          acc[i][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa_[i][e], fb_[q][e], acc[i][q], 0, 0, 0);
  };
  // Cell non-linearities of a finished unit; C/D layout: col = lane & 15, row = 4 * (lane >> 4) + r.  What they read
  // besides the accumulators (bias, old cell state, lengths, old hidden state of rows past their length) is fetched
  // when the stream ENTERS the unit, so the finish itself is arithmetic and stores only.  Nobody else touches these
  // (row, unit) elements during the launch.
  const int e_unit = j0 + wcol * 16 + l15;
  const int e_unit_c = e_unit < H ? e_unit : H - 1;
  float e_bias[4], e_c[8], e_hp[8];
  int e_len[8];
  auto enter_unit = [&](int u) {
    const LstmUnitArgs& L = a.unit[u];
    const int t = a.s - L.t_offset;
    const float* __restrict__ bias = L.bias;
    const float* __restrict__ h_prev = L.h[t & 1];
    const float* __restrict__ cst = L.c;
    const int* __restrict__ lens = a.seq_lengths;
#pragma unroll
    for (int q = 0; q < 4; ++q) e_bias[q] = bias[q * H + e_unit_c];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int row = m0 + wrow * 32 + (e >> 2) * 16 + lq * 4 + (e & 3);
      const int rc = row < B ? row : B - 1;
      const size_t hc = (size_t)rc * H + e_unit_c;
      e_c[e] = cst[hc];
      e_len[e] = lens ? lens[rc] : F;
      e_hp[e] = lens ? h_prev[hc] : 0.f;   // only rows past their length carry the old state over
    }
  };
  auto finish_unit = [&](int u) {
    const LstmUnitArgs& L = a.unit[u];
    const int t = a.s - L.t_offset;
    const bool rev = L.reverse != 0;
    if (e_unit >= H) return;
    float* __restrict__ h_next = L.h[(t + 1) & 1];
    float* __restrict__ cst = L.c;
    float* __restrict__ yout = L.y;
    const long y_ld = L.y_ld, y_col = L.y_col;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int i = e >> 2, r = e & 3;
      const int row = m0 + wrow * 32 + i * 16 + lq * 4 + r;
      if (row >= B) continue;
      const size_t hc = (size_t)row * H + e_unit;
      const bool live = t < e_len[e];
      const int t_out = (rev && live) ? e_len[e] - 1 - t : t;   // a finished reverse row zero-fills the padded slot t
      const float g_i = fsigmoid(acc[i][0][r] + e_bias[0]), g_f = fsigmoid(acc[i][1][r] + e_bias[1]);
      const float g_g = ftanh(acc[i][2][r] + e_bias[2]), g_o = fsigmoid(acc[i][3][r] + e_bias[3]);
      const float c_new = g_f * e_c[e] + g_i * g_g;
      const float h_new = g_o * ftanh(c_new);
      if (live) cst[hc] = c_new;
      h_next[hc] = live ? h_new : e_hp[e];
      if (yout) yout[((size_t)row * F + t_out) * y_ld + y_col + e_unit] = live ? h_new : 0.f;
      if (L.sv_gates) {   // training forward: what back-propagation through time reads
        const size_t rt = (size_t)row * F + t;
        float* sg = L.sv_gates + rt * 4 * H + e_unit;
        sg[0] = g_i; sg[H] = g_f; sg[2 * H] = g_g; sg[3 * H] = g_o;
        L.sv_c[rt * H + e_unit] = live ? c_new : e_c[e];
        if (t + 1 < F) L.sv_hprev[(rt + 1) * H + e_unit] = live ? h_new : e_hp[e];
      }
    }
  };

  LSTM_STAMP(1)
  // ---- prologue: tile 0 -> stage 0, tile 1 in flight
  bool ragged = gload();
  if (ragged) gzero();
  lwrite(lds);
  ld_advance();
  ragged = gload();
  ld_advance();
  __syncthreads();
  fread(lds, 0, fa[0], fb[0]);
  LSTM_STAMP(2)
  int stamp = 3;
  (void)stamp;

  int parity = 0;
  for (int us = 0; us < n_seg; us += 2) {   // one unit = two consecutive segments
    const int unit_tiles = a.seg[seg_beg + us].ntiles + a.seg[seg_beg + us + 1].ntiles;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[i][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    enter_unit(a.seg[seg_beg + us].unit);
    for (int j = 0; j < unit_tiles; ++j) {
      const float* cur = lds + parity * STAGE;
      float* nxt = lds + (parity ^ 1) * STAGE;
      parity ^= 1;
      if (ragged) gzero();     // uniform branch; the registers hold the next tile of the stream
      // ---- chunk 0
      fread(cur, 1, fa[1], fb[1]);
#ifndef LSTM_EXP_NOWRITE
      lwrite(nxt);
#endif
      mma(fa[0], fb[0]);
#pragma unroll
      for (int q = 0; q < 6; ++q) { LSTM_SGB(SG_MFMA, 1); LSTM_SGB(SG_DS_RD, 1); }
#pragma unroll
      for (int q = 0; q < 12; ++q) { LSTM_SGB(SG_MFMA, 2); LSTM_SGB(SG_DS_WR, 1); }
      LSTM_SGB(SG_MFMA, 2);
      // ---- chunk 1
      fread(cur, 2, fa[0], fb[0]);
#ifndef LSTM_EXP_NOLOAD
      ragged = gload();
#endif
      mma(fa[1], fb[1]);
#pragma unroll
      for (int q = 0; q < 6; ++q) { LSTM_SGB(SG_MFMA, 1); LSTM_SGB(SG_DS_RD, 1); }
#pragma unroll
      for (int q = 0; q < 12; ++q) { LSTM_SGB(SG_MFMA, 2); LSTM_SGB(SG_VMEM_RD, 1); }
      LSTM_SGB(SG_MFMA, 2);
      // ---- chunk 2
      fread(cur, 3, fa[1], fb[1]);
      mma(fa[0], fb[0]);
#pragma unroll
      for (int q = 0; q < 6; ++q) { LSTM_SGB(SG_MFMA, 2); LSTM_SGB(SG_DS_RD, 1); }
      LSTM_SGB(SG_MFMA, 20);
#ifndef LSTM_EXP_NOBARRIER
      __syncthreads();
#endif
      // ---- chunk 3
      fread(nxt, 0, fa[0], fb[0]);
      mma(fa[1], fb[1]);
#pragma unroll
      for (int q = 0; q < 6; ++q) { LSTM_SGB(SG_MFMA, 2); LSTM_SGB(SG_DS_RD, 1); }
      LSTM_SGB(SG_MFMA, 20);
      ld_advance();
      LSTM_STAMP(stamp++)
    }
    finish_unit(a.seg[seg_beg + us].unit);
    LSTM_STAMP(stamp++)
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same tile stream over the WHOLE sequence in one cooperative launch (large batches: B > 256; round 3).  A step
// launch of lstm_chain_kernel is the serial path of one workgroup (27 K tiles) plus ~2 us of set-up and the gap to the
// next launch; here a workgroup keeps walking: the stream simply continues with the next wavefront step, its loads two
// tiles ahead as before -- the first tiles of a step are the stored input, which depends on nothing.  What a step needs
// from the other workgroups of its 64-row group (their 32-unit slices of h) is guarded by one counter per (row group,
// unit): each wave adds 1 after its hidden-state stores of a unit have left (two tiles after the unit's finish, when
// the wait for them is free), and a consumer enters a segment that reads such rows once the counter has reached
// workgroups x 4 waves x steps -- its value is fetched a segment ahead, so in the steady state nobody waits.  Hidden states go
// through agent-scope stores / loads (the row group's workgroups sit on different XCDs: plain accesses would see
// their own L2) and through THREE buffers in turn: a workgroup one step ahead then never overwrites rows a slower one
// is still reading (the final state is therefore in buffer F % 3, not F & 1 as after step launches).  Polls are bounded: a counter that never arrives poisons the outputs with NaN instead of hanging
// the GPU.  Same arithmetic in the same order as the step launches: bit-identical results.
// MEASURED (B = 1024, F = 32, 2 x 512): 2.35 ms against 2.07 ms for the 33 step launches (2.24 with plain, L2-cached loads of
// the hidden-state tiles, which would be wrong): the launches follow each other without a gap (kernel time 62.5 us of 62.7
// launch to launch), so there is nothing to win back, and the agent-scope loads / the bookkeeping in the loop cost 4-8 us a
// step.  Opt-in (option lstm_seq, default 0); kept for what it shows.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm_seq_kernel(LstmSeqArgs a) {
  using namespace lc;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  LSTM_STAMP(0)
  const int H = a.H, B = a.B, F = a.F;
  const int j0 = blockIdx.x * BU, m0 = blockIdx.y * BM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wrow = wave >> 1, wcol = wave & 1;
  const int l15 = lane & 15, lq = lane >> 4;
  const int NU = a.n_units, S = F + NU - 1;
  unsigned* __restrict__ cnt = a.counters + (size_t)blockIdx.y * 4;   // this row group's counters, one per unit
  const unsigned n_peers = gridDim.x;                                  // workgroups of a row group
  bool failed = false;

  // ---- per-thread constants
  const int lr = tid >> 4, c16 = tid & 15;                 // global side: row lr + 16 i, 16-byte chunk c16 of the K tile
  int a_row[4], a_len[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = m0 + lr + 16 * i;
    a_row[i] = r < B ? r : B - 1;
    a_len[i] = a.seq_lengths ? a.seq_lengths[a_row[i]] : F;
  }
  int w_row[8];                                            // gate * H + unit of the thread's 8 weight rows
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int wr = lr + 16 * i, un = j0 + (wr & 31);
    w_row[i] = (wr >> 5) * H + (un < H ? un : H - 1);
  }
  const int wr_ofs = lr * BK + ((c16 ^ (lr & 15)) << 2);   // LDS write offset of piece 0 (floats); piece i: + 16 i rows
  const int a_rd = (wrow * 32 + l15) * BK;                 // fragment rows (floats); row tile i: + 16 rows
  const int b_rd = (BM + wcol * 16 + l15) * BK;            // gate g: + 32 rows
  int sw[4];                                               // swizzled chunk offset of k-chunk ch for this lane
#pragma unroll
  for (int ch = 0; ch < 4; ++ch) sw[ch] = ((((ch << 2) ^ (l15 & 12)) | (lq ^ (l15 & 3))) << 2);

  f32x4 g[12];
  bool g_ok = true;
  f32x4 fa[2][2], fb[2][4];
  f32x4 acc[2][4];

  // ---- the load side of the tile stream: current segment (scalar registers), the thread's byte offsets into its two
  // operands (recomputed when the stream enters a segment) and the k tile within the segment
  // segment (s, u, part) of the whole-sequence stream, built here (the step launches get theirs from the host)
  auto make_seg = [&](int s, int u, int part) -> LstmSeg {
    const LstmUnitArgs& L = a.unit[u];
    const int t = s - L.t_offset;
    LstmSeg d;
    d.k = t; d.tstride = 0; d.unit = u; d.pad = 0;
    if (part == 0) {
      if (L.in_from < 0) { d.a = L.in_seq + (size_t)t * L.in_ld; d.lda = F * L.in_ld; }
      else { d.a = a.hs[L.in_from][(t + 1) % 3]; d.lda = H; d.pad = 1 + L.in_from + 8 * (t + 1); }
      d.w = L.w_ih; d.ldw = L.in_k; d.K = L.in_k;
    } else {
      d.a = a.hs[u][t % 3]; d.lda = H; d.w = L.w_hh; d.ldw = H; d.K = H;
      if (t > 0) d.pad = 1 + u + 8 * t;     // (t == 0: the caller's initial state)
    }
    d.ntiles = (d.K + BK - 1) / BK;
    return d;
  };
  // `pad` of a segment whose A rows are hidden states produced inside this launch: 1 + producing unit + 8 * (steps that
  // unit must have published).  The counter is fetched when the stream enters the segment BEFORE (its value is there
  // when it is needed) and only polled if that value was short.
  auto active = [&](int s, int u) -> bool { const int t = s - a.unit[u].t_offset; return t >= 0 && t < F; };
  int ld_s = 0, ld_u = 0, ld_part = 0, ld_kt = 0;
  while (!active(ld_s, ld_u)) { if (++ld_u == NU) { ld_u = 0; ++ld_s; } }
  LstmSeg ld = make_seg(ld_s, ld_u, 0);
  bool ld_end = false, ld_wait = false;
  unsigned dep_seen = 0;     // counter value fetched ahead for the NEXT segment's dependency
  LstmSeg nx = ld;           // the segment after `ld`
  auto peek_next = [&]() {   // nx <- successor of ld (or ld itself at the end of the stream); prefetch its counter
    int s2 = ld_s, u2 = ld_u, p2 = ld_part;
    if (p2 == 0) p2 = 1;
    else {
      p2 = 0;
      do { if (++u2 == NU) { u2 = 0; ++s2; } } while (s2 < S && !active(s2, u2));
    }
    if (s2 >= S) { nx = ld; nx.pad = 0; nx.unit = -1; return; }
    nx = make_seg(s2, u2, p2);
    nx.unit = u2 | (s2 << 8) | (p2 << 30);
  };
  // the counter the NEXT segment depends on, fetched a segment ahead (0 when it depends on nothing: never compared)
  auto prefetch_dep = [&]() {
    dep_seen = nx.pad ? __hip_atomic_load(cnt + ((nx.pad - 1) & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  };
  bool ld_prefetch = false;
  unsigned a_off[4], w_off[8];
  auto seg_offsets = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int tr = a_len[i] - 1 - ld.k;
      a_off[i] = (unsigned)(((long)a_row[i] * ld.lda + (long)(tr > 0 ? tr : 0) * ld.tstride + c16 * 4) * 4);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) w_off[i] = (unsigned)(((long)w_row[i] * ld.ldw + c16 * 4) * 4);
  };
  seg_offsets();
  peek_next();
  prefetch_dep();
  auto gload = [&]() -> bool {   // fetch the tile under the load cursor; returns whether it is ragged
    g_ok = ld_kt * BK + c16 * 4 < ld.K;
    const unsigned back = g_ok ? 0u : (unsigned)(c16 * 16);   // lanes past K re-read chunk 0 of the tile; zeroed later
    gbyte_t pa = (gbyte_t)(ld.a + ld_kt * BK);
    gbyte_t pw = (gbyte_t)(ld.w + ld_kt * BK);
    if (ld.pad) {   // rows other workgroups (other XCDs, other L2s) wrote in this launch: agent-scope loads
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned long long* q = reinterpret_cast<const unsigned long long*>((const char*)pa + (a_off[i] - back));
        const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        g[i] = f32x4{__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)),
                     __uint_as_float((unsigned)hi), __uint_as_float((unsigned)(hi >> 32))};
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) g[i] = *(gvec_t)(pa + (a_off[i] - back));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) g[4 + i] = *(gvec_t)(pw + (w_off[i] - back));
    return (ld_kt + 1) * BK > ld.K;
  };
  auto ld_advance = [&]() {      // uniform; past the end of the stream it parks on a valid tile that is never used
    if (++ld_kt == ld.ntiles) {
      ld_kt = 0;
      if (nx.unit >= 0) {
        ld_s = (nx.unit >> 8) & 0x3fffff; ld_u = nx.unit & 0xff; ld_part = (nx.unit >> 30) & 1;
        ld = nx; ld.unit = ld_u;
        ld_wait = ld.pad != 0;   // hidden states of other workgroups: checked right before the segment's first load
        seg_offsets();
        peek_next();
        ld_prefetch = true;
      } else {
        ld_end = true;
      }
    }
  };
  auto gzero = [&]() {
#pragma unroll
    for (int i = 0; i < 12; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) g[i][e] = g_ok ? g[i][e] : 0.f;
  };
  auto lwrite = [&](float* st) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(st + wr_ofs + i * 16 * BK) = g[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(st + BM * BK + wr_ofs + i * 16 * BK) = g[4 + i];
  };
  auto fread = [&](const float* st, int ch, f32x4 (&fa_)[2], f32x4 (&fb_)[4]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) fa_[i] = *reinterpret_cast<const f32x4*>(st + a_rd + i * 16 * BK + sw[ch]);
#pragma unroll
    for (int q = 0; q < 4; ++q) fb_[q] = *reinterpret_cast<const f32x4*>(st + b_rd + q * 32 * BK + sw[ch]);
  };
  auto mma = [&](const f32x4 (&fa_)[2], const f32x4 (&fb_)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          acc[i][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa_[i][e], fb_[q][e], acc[i][q], 0, 0, 0);
  };
  // Cell non-linearities of a finished unit; C/D layout: col = lane & 15, row = 4 * (lane >> 4) + r.  What they read
  // besides the accumulators (bias, old cell state, lengths, old hidden state of rows past their length) is fetched
  // when the stream ENTERS the unit, so the finish itself is arithmetic and stores only.  Nobody else touches these
  // (row, unit) elements during the launch.
  const int e_unit = j0 + wcol * 16 + l15;
  const int e_unit_c = e_unit < H ? e_unit : H - 1;
  float e_bias[4], e_c[8], e_hp[8];
  int e_len[8];
  auto enter_unit = [&](int u, int t) {
    const LstmUnitArgs& L = a.unit[u];
    const float* __restrict__ bias = L.bias;
    const float* h_prev = a.hs[u][t % 3];
    const float* __restrict__ cst = L.c;
    const int* __restrict__ lens = a.seq_lengths;
#pragma unroll
    for (int q = 0; q < 4; ++q) e_bias[q] = bias[q * H + e_unit_c];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int row = m0 + wrow * 32 + (e >> 2) * 16 + lq * 4 + (e & 3);
      const int rc = row < B ? row : B - 1;
      const size_t hc = (size_t)rc * H + e_unit_c;
      e_c[e] = cst[hc];
      e_len[e] = lens ? lens[rc] : F;
      // only rows past their length carry the old state over (this workgroup's own element, written with an agent-scope
      // store a step ago: read it the same way)
      e_hp[e] = lens ? __hip_atomic_load(h_prev + hc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
    }
  };
  auto finish_unit = [&](int u, int t) {
    const LstmUnitArgs& L = a.unit[u];
    const bool rev = false;
    if (e_unit >= H) return;
    float* h_next = a.hs[u][(t + 1) % 3];
    const float poison = __builtin_nanf("");
    float* __restrict__ cst = L.c;
    float* __restrict__ yout = L.y;
    const long y_ld = L.y_ld, y_col = L.y_col;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int i = e >> 2, r = e & 3;
      const int row = m0 + wrow * 32 + i * 16 + lq * 4 + r;
      if (row >= B) continue;
      const size_t hc = (size_t)row * H + e_unit;
      const bool live = t < e_len[e];
      const int t_out = (rev && live) ? e_len[e] - 1 - t : t;   // a finished reverse row zero-fills the padded slot t
      const float g_i = fsigmoid(acc[i][0][r] + e_bias[0]), g_f = fsigmoid(acc[i][1][r] + e_bias[1]);
      const float g_g = ftanh(acc[i][2][r] + e_bias[2]), g_o = fsigmoid(acc[i][3][r] + e_bias[3]);
      const float c_new = g_f * e_c[e] + g_i * g_g;
      const float h_new = g_o * ftanh(c_new);
      if (live) cst[hc] = failed ? poison : c_new;
      const float h_out = failed ? poison : (live ? h_new : e_hp[e]);
      // other workgroups (on other XCDs) read this row in the next step: a store that reaches the memory side
      __hip_atomic_store(h_next + hc, h_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (yout) yout[((size_t)row * F + t_out) * y_ld + y_col + e_unit] = failed ? poison : (live ? h_new : 0.f);
      if (L.sv_gates) {   // training forward: what back-propagation through time reads
        const size_t rt = (size_t)row * F + t;
        float* sg = L.sv_gates + rt * 4 * H + e_unit;
        sg[0] = g_i; sg[H] = g_f; sg[2 * H] = g_g; sg[3 * H] = g_o;
        L.sv_c[rt * H + e_unit] = live ? c_new : e_c[e];
        if (t + 1 < F) L.sv_hprev[(rt + 1) * H + e_unit] = live ? h_new : e_hp[e];
      }
    }
  };

  LSTM_STAMP(1)
  // ---- prologue: tile 0 -> stage 0, tile 1 in flight
  bool ragged = gload();
  if (ragged) gzero();
  lwrite(lds);
  ld_advance();
  ragged = gload();
  ld_advance();
  __syncthreads();
  fread(lds, 0, fa[0], fb[0]);
  LSTM_STAMP(2)
  int stamp = 3;
  (void)stamp;

  int parity = 0;
  // A unit's new hidden states are PUBLISHED (the row group's counter of that unit goes up by one per workgroup) two
  // tile iterations after its finish: by then every wave's stores have long left (they were issued a whole tile before
  // the loads the first of those iterations waited for), so the wait below costs nothing, and the consumers need them a
  // dozen tiles later at the earliest.
  int pub_unit = -1, pub_wait = 0;
  for (int s = 0; s < S; ++s)
  for (int u = 0; u < NU; ++u) {
    const int t = s - a.unit[u].t_offset;
    if (t < 0 || t >= F) continue;   // idle while the wavefront ramps up or down
    const int unit_tiles = (a.unit[u].in_k + BK - 1) / BK + (H + BK - 1) / BK;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[i][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    enter_unit(u, t);
    for (int j = 0; j < unit_tiles; ++j) {
      const float* cur = lds + parity * STAGE;
      float* nxt = lds + (parity ^ 1) * STAGE;
      parity ^= 1;
      if (ragged) gzero();     // uniform branch; the registers hold the next tile of the stream
      // ---- chunk 0
      fread(cur, 1, fa[1], fb[1]);
      lwrite(nxt);
      mma(fa[0], fb[0]);
#pragma unroll
      for (int q = 0; q < 6; ++q) { LSTM_SGB(SG_MFMA, 1); LSTM_SGB(SG_DS_RD, 1); }
#pragma unroll
      for (int q = 0; q < 12; ++q) { LSTM_SGB(SG_MFMA, 2); LSTM_SGB(SG_DS_WR, 1); }
      LSTM_SGB(SG_MFMA, 2);
      // ---- chunk 1
      fread(cur, 2, fa[0], fb[0]);
      if (pub_unit >= 0 && (--pub_wait == 0 || ld_wait)) {   // (early when this wave is about to wait for the others)
        __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): this wave's hidden-state stores have completed
        if (lane == 0) __hip_atomic_fetch_add(cnt + pub_unit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pub_unit = -1;
      }
      if (ld_wait) {   // (after this wave's own publication: the waves of a row group wait for each other here)
        const unsigned need = n_peers * 4u * (unsigned)(ld.pad >> 3);
        int spins = 0;
        while (dep_seen < need) {
          if (++spins > a.spin_limit) {
            if (!failed && lane == 0) __hip_atomic_fetch_add(a.timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            failed = true;
            break;
          }
          __builtin_amdgcn_s_sleep(2);
          dep_seen = __hip_atomic_load(cnt + ((ld.pad - 1) & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ld_wait = false;
      }
      if (ld_prefetch) { prefetch_dep(); ld_prefetch = false; }   // (after the poll above: one value, one owner)
      ragged = gload();
      mma(fa[1], fb[1]);
#pragma unroll
      for (int q = 0; q < 6; ++q) { LSTM_SGB(SG_MFMA, 1); LSTM_SGB(SG_DS_RD, 1); }
#pragma unroll
      for (int q = 0; q < 12; ++q) { LSTM_SGB(SG_MFMA, 2); LSTM_SGB(SG_VMEM_RD, 1); }
      LSTM_SGB(SG_MFMA, 2);
      // ---- chunk 2
      fread(cur, 3, fa[1], fb[1]);
      mma(fa[0], fb[0]);
#pragma unroll
      for (int q = 0; q < 6; ++q) { LSTM_SGB(SG_MFMA, 2); LSTM_SGB(SG_DS_RD, 1); }
      LSTM_SGB(SG_MFMA, 20);
      __syncthreads();
      // ---- chunk 3
      fread(nxt, 0, fa[0], fb[0]);
      mma(fa[1], fb[1]);
#pragma unroll
      for (int q = 0; q < 6; ++q) { LSTM_SGB(SG_MFMA, 2); LSTM_SGB(SG_DS_RD, 1); }
      LSTM_SGB(SG_MFMA, 20);
      ld_advance();
      LSTM_STAMP(stamp++)
    }
    finish_unit(u, t);
    pub_unit = u; pub_wait = 2;
    LSTM_STAMP(stamp++)
  }
}


// Segment table of one launch: two segments per active unit, grouped by blockIdx.z.
static void lstm_build_chain(LstmWaveArgs& a, int units_per_block) {
  int n = 0, z = 0;
  for (int u0 = 0; u0 < a.n_units; u0 += units_per_block, ++z) {
    a.z_beg[z] = n;
    for (int u = u0; u < u0 + units_per_block && u < a.n_units; ++u) {
      const LstmUnitArgs& L = a.unit[u];
      const int t = a.s - L.t_offset;
      if (t < 0 || t >= a.F) continue;   // idle while the wavefront ramps up or down
      for (int s = 0; s < 2; ++s) {
        LstmSeg d;
        d.k = t; d.tstride = 0; d.unit = u; d.pad = 0;
        if (s == 0) {
          if (L.in_from < 0) {
            d.lda = a.F * L.in_ld;
            if (L.reverse) { d.a = L.in_seq; d.tstride = L.in_ld; }
            else d.a = L.in_seq + (size_t)t * L.in_ld;
          } else {
            d.a = a.unit[L.in_from].h[(t + 1) & 1]; d.lda = a.H;
          }
          d.w = L.w_ih; d.ldw = L.in_k; d.K = L.in_k;
        } else {
          d.a = L.h[t & 1]; d.lda = a.H; d.w = L.w_hh; d.ldw = a.H; d.K = a.H;
        }
        d.ntiles = (d.K + lc::BK - 1) / lc::BK;
        a.seg[n++] = d;
      }
    }
    a.z_cnt[z] = n - a.z_beg[z];
  }
  for (; z < 4; ++z) { a.z_beg[z] = 0; a.z_cnt[z] = 0; }
}

// ---------------------------------------------------------------------------------------------------------------
// Medium batches (17 .. 256 rows: the batched evaluation driver runs chunk c of all recordings as one ragged batch).
// With the 64 x 32 tile of lstm_chain_kernel such a batch is 16 .. 64 workgroups whose waves each walk the whole K of
// their unit (35 us per step however few rows there are).  Here a workgroup owns 32 rows x 16 hidden units of ONE unit
// and its four waves SPLIT K: wave w takes the 16-wide chunk w of every 64-wide K tile (32 MFMAs per tile instead of
// 128), and the partial gate sums meet in LDS at the end, added in wave order (deterministic).  Four times as many
// workgroups, a quarter of the MFMA chain per wave.  Same operands (segment table, two segments per unit), same LDS
// tile layout (XOR-swizzled 16-byte chunks) and fragment order as lstm_chain_kernel; no instruction-level pipelining:
// at this size the launch is latency-bound, two workgroups per CU overlap instead.
// ---------------------------------------------------------------------------------------------------------------
namespace lm {
constexpr int BM = 32, BU = 16, BK = 64;
constexpr int ROWS = BM + 4 * BU;          // 96 staged rows per K tile
constexpr int STAGE = ROWS * BK;
constexpr size_t LDS_BYTES = 2 * (size_t)STAGE * sizeof(float);   // 48 KB; the reduction reuses it (4 x 32 x 64 floats)
}  // namespace lm

__global__ __launch_bounds__(256) void lstm_mid_kernel(LstmWaveArgs a) {
  using namespace lm;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int H = a.H, B = a.B, F = a.F;
  const int j0 = blockIdx.x * BU, m0 = blockIdx.y * BM;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int seg_beg = a.z_beg[blockIdx.z], n_seg = a.z_cnt[blockIdx.z];
  if (n_seg == 0) return;
  const int u = a.seg[seg_beg].unit;
  const LstmUnitArgs& L = a.unit[u];
  const int t = a.seg[seg_beg].k;

  // ---- staging: thread (lr, c16) moves the 16-byte chunk c16 of rows lr + 16 i: i = 0, 1 input rows, 2..5 weight rows
  const int lr = tid >> 4, c16 = tid & 15;
  int a_row[2], a_len[2], w_row[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = m0 + lr + 16 * i;
    a_row[i] = r < B ? r : B - 1;
    a_len[i] = a.seq_lengths ? a.seq_lengths[a_row[i]] : F;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int un = j0 + lr;
    w_row[i] = i * H + (un < H ? un : H - 1);     // gate i, unit j0 + lr
  }
  const int wr_ofs = lr * BK + ((c16 ^ lr) << 2);
  const int sw = ((((wave << 2) ^ (l15 & 12)) | (lq ^ (l15 & 3))) << 2);   // this wave's k chunk, swizzled for row l15

  f32x4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[i][q] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 g[6];
  auto gload = [&](const LstmSeg& sg, int kt) {
    const bool ok = kt * BK + c16 * 4 < sg.K;           // lanes past K stage zeros
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int tr = a_len[i] - 1 - sg.k;
      const float* p = sg.a + (size_t)a_row[i] * sg.lda + (size_t)(tr > 0 ? tr : 0) * sg.tstride + kt * BK + c16 * 4;
      g[i] = ok ? *reinterpret_cast<const f32x4*>(p) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float* p = sg.w + (size_t)w_row[i] * sg.ldw + kt * BK + c16 * 4;
      g[2 + i] = ok ? *reinterpret_cast<const f32x4*>(p) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto lwrite = [&](float* st) {
#pragma unroll
    for (int i = 0; i < 6; ++i) *reinterpret_cast<f32x4*>(st + wr_ofs + i * 16 * BK) = g[i];
  };

  // flat tile stream over the unit's two segments
  const LstmSeg s0 = a.seg[seg_beg], s1 = a.seg[seg_beg + 1];
  const int n_tiles = s0.ntiles + s1.ntiles;
  auto tile_load = [&](int j) { if (j < s0.ntiles) gload(s0, j); else gload(s1, j - s0.ntiles); };

  // what the cell update reads besides the sums: fetched now, used after the K loop.  Wave w finishes the elements
  // e = 2w, 2w + 1 of the 16x16 C/D layout (row = i * 16 + 4 * (lane >> 4) + r with e = 4 i + r, col = lane & 15).
  const int e_unit = j0 + l15, e_unit_c = e_unit < H ? e_unit : H - 1;
  float e_bias[4], e_c[2], e_hp[2];
  int e_len[2], e_rowi[2];
#pragma unroll
  for (int q = 0; q < 4; ++q) e_bias[q] = L.bias[q * H + e_unit_c];
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const int e = 2 * wave + x, row = m0 + (e >> 2) * 16 + lq * 4 + (e & 3);
    const int rc = row < B ? row : B - 1;
    e_rowi[x] = row;
    e_c[x] = L.c[(size_t)rc * H + e_unit_c];
    e_len[x] = a.seq_lengths ? a.seq_lengths[rc] : F;
    e_hp[x] = L.h[t & 1][(size_t)rc * H + e_unit_c];
  }

  tile_load(0);
  lwrite(lds);
  if (n_tiles > 1) tile_load(1);
  __syncthreads();
  for (int j = 0; j < n_tiles; ++j) {
    const float* cur = lds + (j & 1) * STAGE;
    float* nxt = lds + ((j + 1) & 1) * STAGE;
    f32x4 fa[2], fb[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const f32x4*>(cur + (i * 16 + l15) * BK + sw);
#pragma unroll
    for (int q = 0; q < 4; ++q) fb[q] = *reinterpret_cast<const f32x4*>(cur + (BM + q * BU + l15) * BK + sw);
    if (j + 1 < n_tiles) lwrite(nxt);           // tile j + 1 (in registers since the previous iteration)
    if (j + 2 < n_tiles) tile_load(j + 2);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          acc[i][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][e], fb[q][e], acc[i][q], 0, 0, 0);
    __syncthreads();
  }

  // ---- partial sums of the four k chunks -> LDS [wave][element e][gate q][lane], summed in wave order
  float* red = lds;
  static_assert(4 * 8 * 4 * 64 <= 2 * STAGE, "the partial-sum block [4 waves][8][4 gates][64 lanes] reuses the K-tile stages");
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[((wave * 8 + i * 4 + r) * 4 + q) * 64 + lane] = acc[i][q][r];
  __syncthreads();
  if (e_unit >= H) return;
  const bool rev = L.reverse != 0;
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const int e = 2 * wave + x, row = e_rowi[x];
    if (row >= B) continue;
    float gs[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v = red[((0 * 8 + e) * 4 + q) * 64 + lane];
#pragma unroll
      for (int w2 = 1; w2 < 4; ++w2) v += red[((w2 * 8 + e) * 4 + q) * 64 + lane];
      gs[q] = v;
    }
    const size_t hc = (size_t)row * H + e_unit;
    const bool live = t < e_len[x];
    const int t_out = (rev && live) ? e_len[x] - 1 - t : t;
    const float g_i = fsigmoid(gs[0] + e_bias[0]), g_f = fsigmoid(gs[1] + e_bias[1]);
    const float g_g = ftanh(gs[2] + e_bias[2]), g_o = fsigmoid(gs[3] + e_bias[3]);
    const float c_new = g_f * e_c[x] + g_i * g_g;
    const float h_new = g_o * ftanh(c_new);
    if (live) L.c[hc] = c_new;
    L.h[(t + 1) & 1][hc] = live ? h_new : e_hp[x];
    if (L.y) L.y[((size_t)row * F + t_out) * L.y_ld + L.y_col + e_unit] = live ? h_new : 0.f;
    if (L.sv_gates) {   // training forward
      const size_t rt = (size_t)row * F + t;
      float* sg = L.sv_gates + rt * 4 * H + e_unit;
      sg[0] = g_i; sg[H] = g_f; sg[2 * H] = g_g; sg[3 * H] = g_o;
      L.sv_c[rt * H + e_unit] = live ? c_new : e_c[x];
      if (t + 1 < F) L.sv_hprev[(rt + 1) * H + e_unit] = live ? h_new : e_hp[x];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Small batches (the one-recording-at-a-time driver: B = 1, F = 256): a step is a matrix-VECTOR product, bound by
// streaming the weights (13.8 MB per wavefront step for 2 x 512), not by arithmetic; the matrix-core kernel above would
// spend a 64-row tile on one row.  Here a wave owns one hidden unit of one layer: its lanes split K, read the four gate
// rows of the unit as coalesced 16-byte pieces, and reduce with DPP shuffles; 4 units per block, so 2 x 128 blocks
// keep every CU streaming.  Same operands (host-built segment table), same state handling as lstm_chain_kernel.
// ---------------------------------------------------------------------------------------------------------------
// acc + w . v over a lane's four k, in one fixed order (explicit FMAs: both small-batch kernels give the same bits)
__device__ __forceinline__ float dot4_acc(float acc, const f32x4& w, const f32x4& v) {
  return __builtin_fmaf(w[3], v[3], __builtin_fmaf(w[2], v[2], __builtin_fmaf(w[1], v[1], __builtin_fmaf(w[0], v[0], acc))));
}

template <int MB>   // rows handled, B <= MB
__global__ __launch_bounds__(256) void lstm_small_kernel(LstmWaveArgs a) {
  const int n_seg = a.z_cnt[blockIdx.y];
  if (n_seg == 0) return;
  const int seg_beg = a.z_beg[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, B = a.B, F = a.F;
  const int unit = blockIdx.x * 4 + wave;
  if (unit >= H) return;   // no barrier below: whole waves may leave
  const LstmUnitArgs& L = a.unit[a.seg[seg_beg].unit];
  const int t = a.seg[seg_beg].k;
  const int* __restrict__ lens = a.seq_lengths;

  float acc[4][MB];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int b = 0; b < MB; ++b) acc[g][b] = 0.f;

  for (int s = 0; s < 2; ++s) {
    const LstmSeg sg = a.seg[seg_beg + s];
    const float* __restrict__ wbase = sg.w + (size_t)unit * sg.ldw;
    for (int k4 = lane * 4; k4 < sg.K; k4 += 256) {
      f32x4 w[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) w[g] = *reinterpret_cast<const f32x4*>(wbase + (size_t)g * H * sg.ldw + k4);
#pragma unroll
      for (int b = 0; b < MB; ++b) {
        if (b < B) {   // uniform; a predicate, not a `break`: the loop then unrolls completely and acc[][] stays in registers
          const int tr = (lens ? lens[b] : F) - 1 - sg.k;
          const float* row = sg.a + (size_t)b * sg.lda + (size_t)(tr > 0 ? tr : 0) * sg.tstride;
          const f32x4 v = *reinterpret_cast<const f32x4*>(row + k4);
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g][b] = dot4_acc(acc[g][b], w[g], v);
        }
      }
    }
  }
  // wave-wide sums (every lane ends up with all of them)
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int b = 0; b < MB; ++b) {
      if (b < B) {
        float v = acc[g][b];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        acc[g][b] = v;
      }
    }
  // lane b finishes row b
  float gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f;
#pragma unroll
  for (int b = 0; b < MB; ++b)
    if (lane == b) { gi = acc[0][b]; gf = acc[1][b]; gg = acc[2][b]; go = acc[3][b]; }
  if (lane < B) {
    const int row = lane;
    const bool rev = L.reverse != 0;
    const float* __restrict__ bias = L.bias;
    const float* __restrict__ h_prev = L.h[t & 1];
    float* __restrict__ h_next = L.h[(t + 1) & 1];
    float* __restrict__ cst = L.c;
    const size_t hc = (size_t)row * H + unit;
    const int len = lens ? lens[row] : F;
    const bool live = t < len;
    const int t_out = (rev && live) ? len - 1 - t : t;   // a finished reverse row zero-fills the padded slot t
    float h_new = 0.f, c_old = 0.f, c_keep = 0.f, h_carry;
    if (live) {   // (these expressions are kept as they are: lstm_persist_kernel must give the same bits)
      c_old = cst[hc];
      const float c_new = fsigmoid(gf + bias[H + unit]) * c_old + fsigmoid(gi + bias[unit]) * ftanh(gg + bias[2 * H + unit]);
      h_new = fsigmoid(go + bias[3 * H + unit]) * ftanh(c_new);
      cst[hc] = c_new;
      h_next[hc] = h_carry = h_new;
      c_keep = c_new;
    } else {
      h_next[hc] = h_carry = h_prev[hc];
      if (L.sv_gates) c_keep = cst[hc];
    }
    if (L.y) L.y[((size_t)row * F + t_out) * L.y_ld + L.y_col + unit] = h_new;
    if (L.sv_gates) {   // training forward: what back-propagation through time reads
      const size_t rt = (size_t)row * F + t;
      float* sg = L.sv_gates + rt * 4 * H + unit;
      sg[0] = fsigmoid(gi + bias[unit]); sg[H] = fsigmoid(gf + bias[H + unit]);
      sg[2 * H] = ftanh(gg + bias[2 * H + unit]); sg[3 * H] = fsigmoid(go + bias[3 * H + unit]);
      L.sv_c[rt * H + unit] = c_keep;
      if (t + 1 < F) L.sv_hprev[(rt + 1) * H + unit] = h_carry;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// A few rows (4 .. 16: the reference's training batch of 12 windows, the batched evaluation driver's chunk of recordings),
// round 5.  lstm_small_kernel above costs 2.2 us per row and step (34 us per step at 12 rows: a row's 16-byte pieces are
// loaded inside the k loop, 6 lane exchanges per gate sum); the whole-sequence kernel below polls B x H exchange words per
// layer and step in every workgroup (16 us per step at 12 rows).  Here a step is one launch in which a workgroup owns TWO
// hidden units (8 gate columns) and ALL its 256 threads split the unit's K = [input | recurrent] (1024 floats at 2 x 512:
// one 16-byte piece per thread): a thread loads its piece of the B rows and of the 8 weight rows once, 8 B accumulators;
// the sums over a wave's lanes are a reduce-scatter (lane_reduce.h), the four waves' sums meet in LDS and are added in
// wave order; thread (unit, row) applies the cell.  Same state handling and saves as lstm_small_kernel; another summation
// order (option "lstm_fewrows" = 0 selects the kernels above, whose bits the whole-sequence kernel shares).
// ---------------------------------------------------------------------------------------------------------------
template <int MB, int NU>   // rows handled (B <= MB), hidden units per workgroup
__global__ __launch_bounds__(256) void lstm_fewrows_kernel(LstmWaveArgs a) {
  constexpr int C = 4 * NU, NV = C * MB;
  __shared__ float wsum[4][NV + 4];
  const int n_seg = a.z_cnt[blockIdx.y];
  if (n_seg == 0) return;
  const int seg_beg = a.z_beg[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, B = a.B, F = a.F;
  const int u0 = blockIdx.x * NU;
  const LstmUnitArgs& L = a.unit[a.seg[seg_beg].unit];
  const int t = a.seg[seg_beg].k;
  const int* __restrict__ lens = a.seq_lengths;
  // thread (unit f_u, row f_m) finishes a cell: its state and bias are fetched now, under the product
  const int f_u = tid / MB, f_m = tid - f_u * MB;
  const bool fin = tid < NU * MB && f_m < B && u0 + f_u < H;
  const int unit = u0 + (f_u < NU ? f_u : 0);
  float e_bias[4] = {0.f, 0.f, 0.f, 0.f}, e_c = 0.f, e_hp = 0.f;
  int e_len = F;
  if (fin) {
    const size_t hc = (size_t)f_m * H + unit;
#pragma unroll
    for (int q = 0; q < 4; ++q) e_bias[q] = L.bias[q * H + unit];
    e_c = L.c[hc];
    e_hp = L.h[t & 1][hc];
    e_len = lens ? lens[f_m] : F;
  }

  float v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = 0.f;
  {
    const LstmSeg s0 = a.seg[seg_beg], s1 = a.seg[seg_beg + 1];
    const int n0 = s0.K >> 2, n1 = n_seg > 1 ? s1.K >> 2 : 0;       // 16-byte pieces of the two operand segments
    for (int p = tid; p < n0 + n1; p += 256) {
      const bool first = p < n0;
      const int k4 = (first ? p : p - n0) * 4;
      const float* __restrict__ wb = (first ? s0.w : s1.w) + k4;
      const float* __restrict__ ab = (first ? s0.a : s1.a) + k4;
      const int ldw = first ? s0.ldw : s1.ldw, lda = first ? s0.lda : s1.lda;
      const int tstride = first ? s0.tstride : s1.tstride, sk = first ? s0.k : s1.k;
      f32x4 w[C], x[MB];
#pragma unroll
      for (int ul = 0; ul < NU; ++ul)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          w[ul * 4 + q] = *reinterpret_cast<const f32x4*>(wb + (size_t)(q * H + min(u0 + ul, H - 1)) * ldw);
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const int mr = m < B ? m : B - 1;
        const int tr = (lens ? lens[mr] : F) - 1 - sk;
        x[m] = *reinterpret_cast<const f32x4*>(ab + (size_t)mr * lda + (size_t)(tr > 0 ? tr : 0) * tstride);
      }
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int m = 0; m < MB; ++m) v[c * MB + m] = dot4_acc(v[c * MB + m], w[c], x[m]);
    }
  }
  int base = 0, count = 0;
  LaneReduceScatter<NV, 32, NV>::run(v, lane, base, count);
#pragma unroll
  for (int i = 0; i < 4; ++i)      // (count <= 3 for the instantiated sizes; lanes that hold the same sums write the same)
    if (i < count) wsum[wave][base + i] = v[i];
  __syncthreads();
  if (!fin) return;
  float gs[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int o = (f_u * 4 + q) * MB + f_m;
    gs[q] = ((wsum[0][o] + wsum[1][o]) + wsum[2][o]) + wsum[3][o];
  }
  const int row = f_m;
  const bool rev = L.reverse != 0;
  const size_t hc = (size_t)row * H + unit;
  const bool live = t < e_len;
  const int t_out = (rev && live) ? e_len - 1 - t : t;   // a finished reverse row zero-fills the padded slot t
  const float g_i = fsigmoid(gs[0] + e_bias[0]), g_f = fsigmoid(gs[1] + e_bias[1]);
  const float g_g = ftanh(gs[2] + e_bias[2]), g_o = fsigmoid(gs[3] + e_bias[3]);
  const float c_new = g_f * e_c + g_i * g_g;
  const float h_new = g_o * ftanh(c_new);
  if (live) L.c[hc] = c_new;
  L.h[(t + 1) & 1][hc] = live ? h_new : e_hp;
  if (L.y) L.y[((size_t)row * F + t_out) * L.y_ld + L.y_col + unit] = live ? h_new : 0.f;
  if (L.sv_gates) {   // training forward: what back-propagation through time reads
    const size_t rt = (size_t)row * F + t;
    float* sg = L.sv_gates + rt * 4 * H + unit;
    sg[0] = g_i; sg[H] = g_f; sg[2 * H] = g_g; sg[3 * H] = g_o;
    L.sv_c[rt * H + unit] = live ? c_new : e_c;
    if (t + 1 < F) L.sv_hprev[(rt + 1) * H + unit] = live ? h_new : e_hp;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Small batches, whole sequence in ONE launch (the streaming driver: B = 1, F = 256 is 257 dependent wavefront steps).
// The per-step launches above pay a dispatch per step (~6 us back to back) to stream 13.8 MB of weights that never
// change.  Here a wave owns one hidden unit of one layer for the whole sequence and keeps that unit's four gate rows in
// REGISTERS (2 x 512: 4 x 1024 floats over 64 lanes = 64 VGPRs), its lanes b < B keep c and h of the unit; a block is
// four units of one layer and stages the step's input rows ([x_t | h_{t-1}] or [h^{l-1}_t | h^l_{t-1}]) in LDS once.
// What crosses blocks per step is only the new hidden state (B x H floats per layer).  There is no grid barrier: every
// exchanged value is an 8-byte word {value, tag = producing step + 1} written with one agent-scope store, and a consumer
// polls the words it needs until they carry the tag it expects -- the step-to-step critical path is one store reaching
// the memory side plus one load (a counter barrier costs two more round trips: measured 5.9 us per step against 17 us
// with cache-wide release/acquire fences).  Two buffers suffice: a block that overwrites the buffer of step s at step
// s + 2 has consumed every block's step s + 1 output, which those blocks produced after reading step s.
// The kernel is launched cooperatively (all blocks resident), the polls are bounded: a word that never arrives poisons
// the outputs with NaN instead of hanging the GPU.  Same arithmetic, in the same order, as lstm_small_kernel.
// ---------------------------------------------------------------------------------------------------------------
struct LstmPersistArgs {
  LstmUnitArgs unit[4];
  int n_units;
  const int* seq_lengths;
  int B, F, H;
  unsigned long long* xch;   // [2][n_units][B][H] exchange words, zeroed before the launch
  int spin_limit;
  unsigned* timeouts;        // poll_timeout_word()
};

__device__ __forceinline__ unsigned long long xch_pack(float v, unsigned tag) {
  return ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
}

// LDS of the kernel below, one definition for its indexing and its launcher: rows[B][x_t (layer 0's stored input) | h^0 |
// h^1 | ...]
__host__ __device__ constexpr int lstm_persist_row_floats(int in_k0, int n_units, int H) { return in_k0 + n_units * H; }
__host__ __device__ constexpr size_t lstm_persist_lds_floats(int B, int in_k0, int n_units, int H) {
  return (size_t)B * lstm_persist_row_floats(in_k0, n_units, H);
}

template <int P0, int P1>   // 256-wide pieces of the input / recurrent segment (K <= 256 * P)
__global__ __launch_bounds__(256) void lstm_persist_kernel(LstmPersistArgs a) {
  extern __shared__ __attribute__((aligned(16))) float rows[];   // [B][K0 + NL * H]: x_t | h^0 | h^1 | ...
  __shared__ int fail_lds[1];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) fail_lds[0] = 0;
  const int NL = a.n_units, H = a.H, B = a.B, F = a.F;
  // Every block holds units of ALL layers (wave w: layer w % NL): a block that has staged step s + 1 has then consumed
  // the step-s outputs of every layer, which is what makes two exchange buffers enough (see above).
  const int upb = 4 / NL;                       // units per layer per block (a 3-layer stack leaves one wave idle)
  const int l = wave % NL;
  const int unit = blockIdx.x * upb + wave / NL;
  const bool have_unit = wave < upb * NL && unit < H;
  const LstmUnitArgs& U = a.unit[l];
  const int KX = a.unit[0].in_k;              // stored input of layer 0
  const int ldr = lstm_persist_row_floats(KX, NL, H);
  const int K0 = U.in_k, K1 = H;
  const int in_off = l == 0 ? 0 : KX + (l - 1) * H, rec_off = KX + l * H;
  const int* __restrict__ lens = a.seq_lengths;
  const size_t xch_layer = (size_t)B * H, xch_buf = (size_t)NL * xch_layer;

  // this unit's gate rows -> registers
  f32x4 w0[P0][4], w1[P1][4];
#pragma unroll
  for (int p = 0; p < P0; ++p)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int k4 = lane * 4 + p * 256;
      w0[p][g] = (have_unit && k4 < K0) ? *reinterpret_cast<const f32x4*>(U.w_ih + ((size_t)g * H + unit) * K0 + k4)
                                        : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
  for (int p = 0; p < P1; ++p)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int k4 = lane * 4 + p * 256;
      w1[p][g] = (have_unit && k4 < K1) ? *reinterpret_cast<const f32x4*>(U.w_hh + ((size_t)g * H + unit) * K1 + k4)
                                        : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  float bias[4] = {0.f, 0.f, 0.f, 0.f};
  if (have_unit)
#pragma unroll
    for (int g = 0; g < 4; ++g) bias[g] = U.bias[g * H + unit];
  // lane b < B: state of (row b, unit)
  const bool row_lane = have_unit && lane < B;
  const size_t hc = (size_t)lane * H + unit;
  float c_reg = row_lane ? U.c[hc] : 0.f;
  float h_reg = row_lane ? U.h[0][hc] : 0.f;
  const int len = row_lane ? (lens ? lens[lane] : F) : 0;
  bool failed = false;
  __syncthreads();

  const int S = F + NL - 1;
  for (int s = 0; s < S; ++s) {
    const int k = s - l;   // this wave's time step (layer l runs one wavefront step behind layer l - 1)
    // ---- stage the step's rows.  Stored sequence / initial state: plain loads; hidden states produced inside this
    // launch: exchange words of wavefront step s - 1, which carry tag s.
    {
      const unsigned long long* xprev = a.xch + (size_t)((s - 1) & 1) * xch_buf;
      bool bad = false;
      if (s < F)
        for (int i = tid; i < B * KX; i += 256) {
          const int b = i / KX, c = i - b * KX;
          rows[b * ldr + c] = a.unit[0].in_seq[((size_t)b * F + s) * a.unit[0].in_ld + c];
        }
      // Hidden-state slots, all layers in one sweep: "row" q = hl * B + b is the state of layer hl, batch row b, after
      // the layer's step kp = s - 1 - hl, produced at wavefront step s - 1 (kp == -1: the layer starts now, initial
      // state; otherwise unused).  A chunk of polls costs one round trip to the memory side, so a thread keeps RQ rows x
      // P1 words in flight; rows and layers are uniform loop counters, so no per-word index division.
      auto sweep = [&](auto rq_tag) {
        constexpr int RQ = decltype(rq_tag)::value;
        const int n_rows = NL * B;
        for (int q0 = 0; q0 < n_rows; q0 += RQ) {
          unsigned long long wv[RQ][P1];
          int hl = q0 / B, b = q0 - hl * B;   // uniform
#pragma unroll
          for (int u = 0; u < RQ; ++u) {
            const bool valid = q0 + u < n_rows;
            const int kp = s - 1 - hl;
#pragma unroll
            for (int p = 0; p < P1; ++p) {
              const int j = tid + p * 256;
              wv[u][p] = (unsigned long long)(unsigned)s << 32;     // "nothing to wait for"
              if (valid && j < H) {
                const size_t m = ((size_t)hl * B + b) * H + j;
                if (kp >= 0 && kp < F) wv[u][p] = __hip_atomic_load(xprev + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else if (kp == -1) wv[u][p] |= __float_as_uint(a.unit[hl].h[0][(size_t)b * H + j]);
              }
            }
            if (++b == B) { b = 0; ++hl; }
          }
          hl = q0 / B; b = q0 - hl * B;
#pragma unroll
          for (int u = 0; u < RQ; ++u) {
            const bool valid = q0 + u < n_rows;
#pragma unroll
            for (int p = 0; p < P1; ++p) {
              const int j = tid + p * 256;
              if (valid && j < H) {
                const size_t m = ((size_t)hl * B + b) * H + j;
                int spins = 0;
                while ((unsigned)(wv[u][p] >> 32) != (unsigned)s) {
                  if (++spins > a.spin_limit) { bad = true; break; }
                  __builtin_amdgcn_s_sleep(1);
                  wv[u][p] = __hip_atomic_load(xprev + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                rows[b * ldr + KX + hl * H + j] = __uint_as_float((unsigned)wv[u][p]);
              }
            }
            if (++b == B) { b = 0; ++hl; }
          }
        }
      };
      if (NL * B <= 4) sweep(std::integral_constant<int, 4>{});
      else if (NL * B <= 8) sweep(std::integral_constant<int, 8>{});
      else sweep(std::integral_constant<int, 16>{});
      if (bad) fail_lds[0] = 1;
    }
    __syncthreads();
    if (fail_lds[0]) {
      if (!failed && tid == 0) __hip_atomic_fetch_add(a.timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      failed = true;
    }
    if (have_unit && k >= 0 && k < F) {
      for (int b0 = 0; b0 < B; b0 += 4) {
        float acc[4][4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[g][b] = 0.f;
#pragma unroll
        for (int p = 0; p < P0; ++p) {
          const int k4 = lane * 4 + p * 256;
          if (k4 < K0)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
              if (b0 + b >= B) break;   // uniform
              const f32x4 v = *reinterpret_cast<const f32x4*>(rows + (b0 + b) * ldr + in_off + k4);
#pragma unroll
              for (int g = 0; g < 4; ++g) acc[g][b] = dot4_acc(acc[g][b], w0[p][g], v);
            }
        }
#pragma unroll
        for (int p = 0; p < P1; ++p) {
          const int k4 = lane * 4 + p * 256;
          if (k4 < K1)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
              if (b0 + b >= B) break;
              const f32x4 v = *reinterpret_cast<const f32x4*>(rows + (b0 + b) * ldr + rec_off + k4);
#pragma unroll
              for (int g = 0; g < 4; ++g) acc[g][b] = dot4_acc(acc[g][b], w1[p][g], v);
            }
        }
        // Wave-wide sums of the 16 (gate, row) partials as a reduce-scatter: at offset 32 a lane keeps 8 of them, at 16
        // four, ... so 8 + 4 + 2 + 1 + 2 shuffles instead of 16 x 6.  The pairs added at each offset are the ones of the
        // xor butterfly of lstm_small_kernel, so the sums have the same bits.
        float r8[8], r4[4], r2[2], r1;
        {
          const bool hi32 = (lane & 32) != 0, hi16 = (lane & 16) != 0, hi8 = (lane & 8) != 0, hi4 = (lane & 4) != 0;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float lo = acc[j >> 2][j & 3], hi = acc[2 + (j >> 2)][j & 3];   // values j and j + 8 (index g * 4 + b)
            r8[j] = (hi32 ? hi : lo) + __shfl_xor(hi32 ? lo : hi, 32, 64);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) r4[j] = (hi16 ? r8[j + 4] : r8[j]) + __shfl_xor(hi16 ? r8[j] : r8[j + 4], 16, 64);
#pragma unroll
          for (int j = 0; j < 2; ++j) r2[j] = (hi8 ? r4[j + 2] : r4[j]) + __shfl_xor(hi8 ? r4[j] : r4[j + 2], 8, 64);
          r1 = (hi4 ? r2[1] : r2[0]) + __shfl_xor(hi4 ? r2[0] : r2[1], 4, 64);
          r1 += __shfl_xor(r1, 2, 64);
          r1 += __shfl_xor(r1, 1, 64);
        }
        // lane 4 * q (and its quad) now holds value q' with bits (q >> 3, q >> 2 & 1, q >> 1 & 1, q & 1) = (hi32, hi16,
        // hi8, hi4) of the lane, i.e. value index = lane >> 2 read as g * 4 + b.  Row b's lane collects its four gates.
        float gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int r1i = __float_as_int(r1);
          const float vi = __int_as_float(__builtin_amdgcn_readlane(r1i, 4 * (0 * 4 + b)));
          const float vf = __int_as_float(__builtin_amdgcn_readlane(r1i, 4 * (1 * 4 + b)));
          const float vg = __int_as_float(__builtin_amdgcn_readlane(r1i, 4 * (2 * 4 + b)));
          const float vo = __int_as_float(__builtin_amdgcn_readlane(r1i, 4 * (3 * 4 + b)));
          if (lane == b0 + b) { gi = vi; gf = vf; gg = vg; go = vo; }
        }
        if (lane >= b0 && lane < b0 + 4 && lane < B) {   // lane b finishes row b
          const bool live = k < len;
          float h_new = 0.f;
          if (live) {
            const float c_new = fsigmoid(gf + bias[1]) * c_reg + fsigmoid(gi + bias[0]) * ftanh(gg + bias[2]);
            h_new = fsigmoid(go + bias[3]) * ftanh(c_new);
            c_reg = c_new;
            h_reg = h_new;
          }
          const float poison = __builtin_nanf("");
          __hip_atomic_store(a.xch + (size_t)(s & 1) * xch_buf + (size_t)l * xch_layer + hc,
                             xch_pack(failed ? poison : h_reg, (unsigned)(s + 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (U.y) U.y[((size_t)lane * F + k) * U.y_ld + U.y_col + unit] = failed ? poison : h_new;
          if (U.sv_gates) {   // training forward: what back-propagation through time reads (as lstm_small_kernel)
            const size_t rt = (size_t)lane * F + k;
            float* sg = U.sv_gates + rt * 4 * H + unit;
            sg[0] = fsigmoid(gi + bias[0]); sg[H] = fsigmoid(gf + bias[1]);
            sg[2 * H] = ftanh(gg + bias[2]); sg[3 * H] = fsigmoid(go + bias[3]);
            U.sv_c[rt * H + unit] = failed ? poison : c_reg;
            if (k + 1 < F) U.sv_hprev[(rt + 1) * H + unit] = failed ? poison : h_reg;
          }
        }
      }
    }
    __syncthreads();   // the rows are re-staged next step
  }
  if (row_lane) {
    const float poison = __builtin_nanf("");
    U.c[hc] = failed ? poison : c_reg;
    U.h[F & 1][hc] = failed ? poison : h_reg;   // where the step-by-step kernels leave the final state
  }
}

template <int P0, int P1>
static hipError_t launch_lstm_persist_cfg(LstmPersistArgs& a, size_t lds, int grid, hipStream_t stream, bool* fits) {
  const void* fn = reinterpret_cast<const void*>(lstm_persist_kernel<P0, P1>);
  int capacity = 0;
  if (hipError_t e = coresident_blocks(fn, 256, lds, &capacity)) return e;
  *fits = grid <= capacity;
  if (!*fits) return hipSuccess;
  void* params[] = {&a};
  if (hipLaunchCooperativeKernel(fn, dim3(grid), dim3(256), params, (unsigned)lds, stream) != hipSuccess) {
    (void)hipGetLastError();   // no cooperative launch in this context: the caller steps launch by launch
    *fits = false;
  }
  return hipSuccess;
}

size_t lstm_persist_xch_floats(int n_units, int B, int H) { return (size_t)2 * n_units * B * H * 2; }

// Whole-sequence launch for a stacked uni-directional LSTM on a small batch; *done = false when the configuration is
// outside what the kernel covers (the caller then steps the wavefront launch by launch).
hipError_t launch_lstm_persist(const LstmWaveArgs& w, float* xch, hipStream_t stream, bool* done) {
  *done = false;
  if (w.B > LSTM_PERSIST_B || w.H % 4 != 0 || w.H > 512 || w.n_units < 1 || w.n_units > 4) return hipSuccess;
  // from 4 rows on a step launch of lstm_fewrows_kernel is faster than polling B x H exchange words per layer and step
  if (options().lstm_fewrows != 0 && w.B >= LSTM_FEWROWS_MIN_B) return hipSuccess;
  // (a block's four waves cover all layers: 4, 2 or 1 units of each)
  int k0max = 0;
  for (int u = 0; u < w.n_units; ++u) {
    const LstmUnitArgs& U = w.unit[u];
    if (U.reverse || U.in_k % 4 != 0 || U.in_k > 512 || U.t_offset != u) return hipSuccess;
    if (u == 0 ? U.in_from >= 0 : U.in_from != u - 1) return hipSuccess;
    k0max = U.in_k > k0max ? U.in_k : k0max;
  }
  LstmPersistArgs a;
  for (int u = 0; u < 4; ++u) a.unit[u] = w.unit[u < w.n_units ? u : 0];
  a.n_units = w.n_units; a.seq_lengths = w.seq_lengths; a.B = w.B; a.F = w.F; a.H = w.H;
  a.xch = reinterpret_cast<unsigned long long*>(xch);
  a.spin_limit = options().spin_limit > 0 ? options().spin_limit : 1 << 20;
  a.timeouts = poll_timeout_word();
  if (!a.timeouts) return hipErrorOutOfMemory;
  const size_t lds = lstm_persist_lds_floats(w.B, w.unit[0].in_k, w.n_units, w.H) * sizeof(float);
  const int upb = 4 / w.n_units;
  const int grid = (w.H + upb - 1) / upb;
  if (lds > 128 * 1024) return hipSuccess;
  hipError_t e = hipMemsetAsync(xch, 0, lstm_persist_xch_floats(w.n_units, w.B, w.H) * sizeof(float), stream);
  if (e != hipSuccess) return e;
  const bool wide_in = k0max > 256, wide_h = w.H > 256;
  if (wide_in && wide_h) e = launch_lstm_persist_cfg<2, 2>(a, lds, grid, stream, done);
  else if (wide_h) e = launch_lstm_persist_cfg<1, 2>(a, lds, grid, stream, done);
  else if (wide_in) e = launch_lstm_persist_cfg<2, 1>(a, lds, grid, stream, done);
  else e = launch_lstm_persist_cfg<1, 1>(a, lds, grid, stream, done);
  return e;
}

constexpr int LSTM_SMALL_B = 16;
// lstm_mid_kernel against lstm_chain_kernel, us per wavefront step of the 2 x 512 stack (scripts/dev/bench_lstm_mid.py):
// B = 17..128: 17 vs 36, B = 256: 22 vs 37, B = 512: 44 vs 37.
constexpr int LSTM_MID_B = 256;

hipError_t launch_lstm_wave(const LstmWaveArgs& a_in, hipStream_t stream) {
  LstmWaveArgs a = a_in;
  if (a.B <= LSTM_SMALL_B) {   // weight-streaming matrix-vector kernel, one z slice per unit
    lstm_build_chain(a, 1);
    if (options().lstm_fewrows != 0 && a.B >= LSTM_FEWROWS_MIN_B && a.H % 2 == 0) {   // all threads split K, reduce-scatter
      bool ok = true;
      for (int z = 0; z < a.n_units; ++z)
        for (int i = 0; i < a.z_cnt[z]; ++i) ok = ok && a.seg[a.z_beg[z] + i].K % 4 == 0 && a.z_cnt[z] <= 2;
      if (ok) {
        if (a.B <= 4) hipLaunchKernelGGL((lstm_fewrows_kernel<4, 2>), dim3(a.H / 2, a.n_units), dim3(256), 0, stream, a);
        else if (a.B <= 8) hipLaunchKernelGGL((lstm_fewrows_kernel<8, 2>), dim3(a.H / 2, a.n_units), dim3(256), 0, stream, a);
        else if (a.B <= 12) hipLaunchKernelGGL((lstm_fewrows_kernel<12, 2>), dim3(a.H / 2, a.n_units), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((lstm_fewrows_kernel<16, 1>), dim3(a.H, a.n_units), dim3(256), 0, stream, a);
        return hipGetLastError();
      }
    }
    dim3 grid((a.H + 3) / 4, a.n_units);
    if (a.B <= 4) hipLaunchKernelGGL(lstm_small_kernel<4>, grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(lstm_small_kernel<LSTM_SMALL_B>, grid, dim3(256), 0, stream, a);
    return hipGetLastError();
  }
  if (a.B <= LSTM_MID_B) {   // K split over the waves of 32 x 16 tiles, one z slice per unit
    lstm_build_chain(a, 1);
    if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(lstm_mid_kernel), lm::LDS_BYTES)) return e;
    dim3 grid((a.H + lm::BU - 1) / lm::BU, (a.B + lm::BM - 1) / lm::BM, a.n_units);
    hipLaunchKernelGGL(lstm_mid_kernel, grid, dim3(256), lm::LDS_BYTES, stream, a);
    return hipGetLastError();
  }
  const int tiles = ((a.H + lc::BU - 1) / lc::BU) * ((a.B + lc::BM - 1) / lc::BM);
  // Chain all units in one block (equal work per block) once the tiles alone fill the CUs; spread them otherwise.
  const int units_per_block = tiles >= 192 ? a.n_units : 1;
  lstm_build_chain(a, units_per_block);
  if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(lstm_chain_kernel), lc::LDS_BYTES)) return e;
  dim3 grid((a.H + lc::BU - 1) / lc::BU, (a.B + lc::BM - 1) / lc::BM,
            (a.n_units + units_per_block - 1) / units_per_block);
  hipLaunchKernelGGL(lstm_chain_kernel, grid, dim3(256), lc::LDS_BYTES, stream, a);
  return hipGetLastError();
}

size_t lstm_seq_counter_uints(int B) { return (size_t)((B + lc::BM - 1) / lc::BM) * 4; }

// Whole sequence for a stacked uni-directional LSTM on a large batch; *done = false when the configuration is outside
// what the kernel covers or its workgroups cannot all be resident (the caller then steps launch by launch).
hipError_t launch_lstm_seq(const LstmWaveArgs& w, float* const* h_third, unsigned* counters, hipStream_t stream, bool* done) {
  *done = false;
  if (w.B <= LSTM_MID_B || w.n_units < 1 || w.n_units > 4 || !counters) return hipSuccess;
  for (int u = 0; u < w.n_units; ++u) {
    const LstmUnitArgs& U = w.unit[u];
    if (U.reverse || U.t_offset != u || (u == 0 ? U.in_from >= 0 : U.in_from != u - 1) || !h_third[u]) return hipSuccess;
    // a unit's hidden states are published at most two tile iterations after its finish, inside the NEXT unit's stream:
    // every unit needs a few tiles for that to precede the next finish (the released sizes have 11 and 16)
    if ((U.in_k + lc::BK - 1) / lc::BK + (w.H + lc::BK - 1) / lc::BK < 4) return hipSuccess;
  }
  LstmSeqArgs a;
  for (int u = 0; u < 4; ++u) {
    a.unit[u] = w.unit[u < w.n_units ? u : 0];
    const int uu = u < w.n_units ? u : 0;
    a.hs[u][0] = w.unit[uu].h[0]; a.hs[u][1] = w.unit[uu].h[1]; a.hs[u][2] = h_third[uu];
  }
  a.n_units = w.n_units; a.seq_lengths = w.seq_lengths; a.B = w.B; a.F = w.F; a.H = w.H;
  a.counters = counters;
  a.spin_limit = options().spin_limit > 0 ? options().spin_limit : 1 << 16;   // ~0.1 s per poll at most: a lost counter
  a.timeouts = poll_timeout_word();                                            // poisons and is counted, it does not hang
  if (!a.timeouts) return hipErrorOutOfMemory;
  const dim3 grid((w.H + lc::BU - 1) / lc::BU, (w.B + lc::BM - 1) / lc::BM);
  const void* fn = reinterpret_cast<const void*>(lstm_seq_kernel);
  int capacity = 0;
  if (hipError_t e = coresident_blocks(fn, 256, lc::LDS_BYTES, &capacity)) return e;
  if ((int)(grid.x * grid.y) > capacity) return hipSuccess;
  hipError_t e = hipMemsetAsync(counters, 0, lstm_seq_counter_uints(w.B) * sizeof(unsigned), stream);
  if (e != hipSuccess) return e;
  void* params[] = {&a};
  if (hipLaunchCooperativeKernel(fn, grid, dim3(256), params, (unsigned)lc::LDS_BYTES, stream) != hipSuccess) {
    (void)hipGetLastError();   // no cooperative launch in this context: the caller steps launch by launch
    return hipSuccess;
  }
  *done = true;
  return hipSuccess;
}

}  // namespace empose
