// The whole sequence of a stacked uni-directional LSTM on a MEDIUM batch (4 .. 64 rows: the chunks of the batched evaluation
// driver, scripts/evaluate_real.py, BASELINE configs[3]; reference nn/layers.py:133-157) in ONE cooperative launch, on three
// bf16 pieces per operand (bf16x3.h).  Round 6.
//
// Why: a step launch of lstm_mid_x3_kernel takes 8.4 us at 32 rows (profiles/r06i_lstm_mid_ring_lab.txt): 4.8 us are there
// without its K loop (dispatch, the prologue's state loads, the finish), and the K loop is 3.6 us because every launch
// streams the 25 MB of weight pieces of the two layers again -- the L2 does not keep them across launches -- which is what
// the memory side delivers, however many fragments are in flight.  256 frames x 14 chunks are 3600 such steps of a 40 ms pass.
//
// Here a workgroup owns the same tile as in lstm_mid_x3.hip -- 8 hidden units x 4 gates = one 32-column tile of one layer,
// up to 64 rows, its four waves splitting K (wave w: k-steps w, w + 4, ...) -- for ALL steps, and
//   * keeps its weight fragments in REGISTERS (16 k-steps x 3 pieces x 4 registers = 192 of the wave's 512): a step reads
//     only the A planes, 0.4 MB shared by the 16 workgroups of an XCD through its L2;
//   * keeps cell and hidden state of its (row, unit) cells in registers;
//   * hands the new hidden state over as A planes at an address no step has used before -- slot t + 1 of the layer's
//     [F + 1] sets of planes -- with write-through stores, waits for their acknowledgement, then raises its progress counter
//     (one agent-scope store); a consumer polls the 64 counters of its own layer and of the layer below (one wave-wide load
//     each) and then reads the planes with ordinary loads: no line of a fresh slot can be stale in any cache, so the planes go
//     through the L2 like any operand.  (Tagged words polled by every consumer, as in lstm_persist_kernel, would take every
//     workgroup's copy of the planes from the memory side: 24 MB per step, the traffic this kernel is there to remove.)
// MEASURED (profiles/r06i_lstm_midseq_lab.txt, 2 x 512, 32 rows): 10.5 us per step against 8.2 us for the step launches, so
// the kernel is OPT-IN (option lstm_midseq = 1).  With the hand-over compiled out a step is 5.9 us (products 2 us, the rest
// LDS exchange, cell update and three barriers); reading planes that another XCD has just written adds 1.6 us (one trip to
// the memory side), waiting for the store acknowledgement 0.3 - 0.9 us and the counters 2.7 us (their own trip plus the skew
// of 128 workgroups): a hand-over between XCDs is three dependent trips to the memory side, a kernel boundary costs less.
// Same products, in the same order, with the same two alternating accumulators per row tile as lstm_mid_x3_kernel: the
// outputs have the same bits (tests/test_hip_round6.py).  One workgroup per CU, one wave per SIMD (bf16x3.h); stores are
// issued in the finish only, when no MFMA of the workgroup is in flight.  The polls are bounded: a counter that never
// arrives poisons the outputs with NaN and counts itself in the poll-timeout word (kernels.h) instead of hanging the GPU.
#include "bf16x3.h"
#include "gemm_epilogue.h"

namespace empose {

namespace lq3 {
constexpr int BM = 64, BU = 8, NT = 256;
constexpr int PLD = BM + 4;
constexpr int PART_FLOATS = 4 * 32 * PLD;
constexpr int HX_FLOATS = BM * (BU + 1);
constexpr size_t LDS_BYTES = 84 * 1024;                       // > half a CU: one workgroup per CU
static_assert((PART_FLOATS + HX_FLOATS + 4) * 4 <= (int)LDS_BYTES, "LDS layout");
constexpr int FRAG = 512;
constexpr int MAXW = 16;                                      // k-steps a wave keeps in registers: K <= 4 * 16 * 16
}  // namespace lq3

typedef const __attribute__((address_space(1))) u32x4_t* lq3_gvec_t;
typedef const __attribute__((address_space(1))) unsigned short* lq3_gptr_t;

__device__ __forceinline__ float lq3_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float lq3_tanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }

// 16 bytes to the memory side (agent scope: what `__hip_atomic_store(..., __HIP_MEMORY_SCOPE_AGENT)` emits for 8)
__device__ __forceinline__ void lq3_store_through(unsigned short* p, u32x4_t v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}

template <int RTS, int D>   // row tiles of 32 (B <= 32 * RTS), A fragments in flight
__global__ __launch_bounds__(lq3::NT) void lstm_midseq_x3_kernel(LstmMidSeqArgs a) {
  X3_EXCLUSIVE_SIMD();
  using namespace lq3;
  extern __shared__ __attribute__((aligned(16))) float part[];
  float* hx = part + PART_FLOATS;
  int* fail_lds = reinterpret_cast<int*>(hx + HX_FLOATS);
  const int H = a.H, B = a.B, F = a.F, NL = a.n_units;
  const int jb = blockIdx.x, JB = H / BU, j0 = jb * BU;
  const int l = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const LstmMidSeqUnit& U = a.unit[l];
  const int KS_h = H / 16, KS_in = U.ks_in, KS = KS_in + KS_h;
  const int RT = (B + 31) / 32;
  const size_t plane = (size_t)RT * KS_h * 3 * FRAG;            // bf16 elements of one slot of hidden-state planes
  if (tid == 0) fail_lds[0] = 0;

  // ---- this wave's weight fragments, for the whole sequence
  const int n_w = (KS - wave + 3) / 4;
  u32x4_t W[MAXW][3];
#pragma unroll
  for (int i = 0; i < MAXW; ++i) {
    const int g = wave + 4 * i;
    if (i < n_w) {
      const bool in = g < KS_in;
      const int ks = in ? g : g - KS_in;
      lq3_gptr_t wb = (lq3_gptr_t)(in ? U.w3_ih : U.w3_hh) + (((size_t)ks * JB + jb) * 3) * FRAG + lane * 8;
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) W[i][pc] = *(lq3_gvec_t)(wb + pc * FRAG);
    } else {
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) W[i][pc] = u32x4_t{0u, 0u, 0u, 0u};
    }
  }

  // ---- the finishing thread's cells: row f_row, units j0 + 2 f_up, + 1; their state lives in registers
  const int f_row = tid & 63, f_up = tid >> 6;
  const int g_row = f_row, g_rowc = g_row < B ? g_row : B - 1;
  const int g_unit = j0 + 2 * f_up;
  const bool row_used = (RTS == 2 || f_row < 32) && g_row < B;
  const int e_len = a.seq_lengths ? a.seq_lengths[g_rowc] : F;
  float c_reg[2], h_reg[2], e_bias[4][2];
  {
    const size_t hc = (size_t)g_rowc * H + g_unit;
    c_reg[0] = U.c[hc]; c_reg[1] = U.c[hc + 1];
    h_reg[0] = U.h0[hc]; h_reg[1] = U.h0[hc + 1];
#pragma unroll
    for (int q = 0; q < 4; ++q) { e_bias[q][0] = U.bias[q * H + g_unit]; e_bias[q][1] = U.bias[q * H + g_unit + 1]; }
  }
  const unsigned* f_own = a.flags + (size_t)l * JB;
  const unsigned* f_below = a.flags + (size_t)(l > 0 ? l - 1 : 0) * JB;
  unsigned* f_mine = a.flags + (size_t)l * JB + jb;
  bool failed = false;
  __syncthreads();

  const int S = F + NL - 1;
  for (int s = 0; s < S; ++s) {
    const int t = s - l;                       // this layer's time step (layer l runs l wavefront steps behind layer 0)
    const bool active = t >= 0 && t < F;       // (uniform)
    if (active) {
      // ---- every workgroup of this layer and of the layer below has finished wavefront step s - 1
#ifndef LQ3_LAB_NOPOLL   // (dev, scripts/dev/lstm_midseq_lab.sh: parts of a step compiled out -- results are then wrong)
      if (s > 0 && !failed) {       // (a workgroup that gave up once does not wait again: it only poisons and counts on)
        int spins = 0;
        for (;;) {
          bool ok = true;
          for (int i = lane; i < JB; i += 64) {
            ok = ok && __hip_atomic_load(f_own + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)s;
            if (l > 0) ok = ok && __hip_atomic_load(f_below + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)s;
          }
          if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
          if (++spins > a.spin_limit) { fail_lds[0] = 1; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");       // (the planes are read after the counters, in program order)
      }
#endif
      // ---- the step's A planes: input = the stored sequence (layer 0) or slot t + 1 of the layer below; recurrent = slot t
#ifdef LQ3_LAB_SAMEPLANES   // every step reads slot 0 / time 0 again: the loads hit the caches
      const unsigned short* const p_in = l == 0 ? U.in3 : a.unit[l - 1].xa;
      const unsigned short* const p_rec = U.xa;
#else
      const unsigned short* const p_in = l == 0 ? U.in3 + (size_t)t * U.in_t_stride
                                                : a.unit[l - 1].xa + (size_t)(t + 1) * plane;
      const unsigned short* const p_rec = U.xa + (size_t)t * plane;
#endif
      f32x16 acc[RTS][2];
#pragma unroll
      for (int r = 0; r < RTS; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int v = 0; v < 16; ++v) acc[r][h][v] = 0.f;
      u32x4_t fa[D][RTS][3];
      auto load = [&, p_in, p_rec](u32x4_t (&A)[RTS][3], int i) {
        const int g = wave + 4 * i;
        const bool in = g < KS_in;
        const int ks = in ? g : g - KS_in, ksn = in ? KS_in : KS_h;
        lq3_gptr_t ab = (lq3_gptr_t)(in ? p_in : p_rec) + ((size_t)ks * 3) * FRAG + lane * 8;
        const size_t rt_stride = RT > 1 ? (size_t)ksn * 3 * FRAG : 0;
#pragma unroll
        for (int r = 0; r < RTS; ++r)
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) A[r][pc] = *(lq3_gvec_t)(ab + r * rt_stride + pc * FRAG);
      };
      auto mma = [&](const u32x4_t (&A)[RTS][3], const u32x4_t (&Wf)[3]) {
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
          for (int r = 0; r < RTS; ++r)
            acc[r][p & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, A[r][X3_PA[p]]),
                                                                    __builtin_bit_cast(bf16x8_t, Wf[X3_PB[p]]), acc[r][p & 1], 0, 0, 0);
      };
#pragma unroll
      for (int d = 0; d < D; ++d)
        if (d < n_w) load(fa[d], d);
#pragma unroll
      for (int i = 0; i < MAXW; ++i) {
        if (i < n_w) mma(fa[i % D], W[i]);                       // (uniform)
        if (i + D < MAXW) { if (i + D < n_w) load(fa[i % D], i + D); }
      }
      // ---- partial sums -> LDS as [wave][column][row]
      {
        float* pw = part + (size_t)wave * 32 * PLD;
#pragma unroll
        for (int r = 0; r < RTS; ++r)
#pragma unroll
          for (int v = 0; v < 4; ++v)
            *reinterpret_cast<f32x4*>(pw + l31 * PLD + r * 32 + 8 * v + 4 * lh) =
                f32x4{acc[r][0][4 * v] + acc[r][1][4 * v], acc[r][0][4 * v + 1] + acc[r][1][4 * v + 1],
                      acc[r][0][4 * v + 2] + acc[r][1][4 * v + 2], acc[r][0][4 * v + 3] + acc[r][1][4 * v + 3]};
      }
      __syncthreads();
      if (fail_lds[0]) {
        if (!failed && tid == 0) __hip_atomic_fetch_add(a.timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        failed = true;
      }
      // ---- finish: thread (row, 2 units); column of gate q of unit u: q * 8 + u
      const float poison = __builtin_nanf("");
      const bool live = t < e_len;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        float gsum[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float* ps = part + (q * BU + 2 * f_up + e) * PLD + f_row;
          gsum[q] = (RTS == 2 || f_row < 32) ? ((ps[0] + ps[32 * PLD]) + ps[2 * 32 * PLD]) + ps[3 * 32 * PLD] : 0.f;
        }
        const float g_i = lq3_sigmoid(gsum[0] + e_bias[0][e]), g_f = lq3_sigmoid(gsum[1] + e_bias[1][e]);
        const float g_g = lq3_tanh(gsum[2] + e_bias[2][e]), g_o = lq3_sigmoid(gsum[3] + e_bias[3][e]);
        const float c_new = g_f * c_reg[e] + g_i * g_g;
        const float h_new = g_o * lq3_tanh(c_new);
        if (live) { c_reg[e] = c_new; h_reg[e] = h_new; }
        else if (!a.seq_lengths) h_reg[e] = 0.f;
        if (failed) { c_reg[e] = poison; h_reg[e] = poison; }
        if (row_used && U.y)
          U.y[((size_t)g_row * F + t) * U.y_ld + U.y_col + g_unit + e] = failed ? poison : (live ? h_new : 0.f);
        hx[f_row * (BU + 1) + 2 * f_up + e] = h_reg[e];
      }
      __syncthreads();
      // the new hidden values as pieces, slot t + 1 of this layer: where step s + 1 of this layer and of the layer above read
      if (tid < BM && tid < B && (RTS == 2 || tid < 32)) {
        const float* src = hx + tid * (BU + 1);
        const Pieces q = split8(src[0], src[1], src[2], src[3], src[4], src[5], src[6], src[7]);
        const int ks = j0 >> 4, ln = (tid & 31) + 32 * ((j0 & 15) >> 3);
        unsigned short* o = U.xa + (size_t)(t + 1) * plane + ((((size_t)(tid >> 5)) * KS_h + ks) * 3) * FRAG + ln * 8;
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) lq3_store_through(o + pc * FRAG, q.p[pc]);
      }
#ifndef LQ3_LAB_NOACK
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the stores have been acknowledged by the memory side
#endif
      __syncthreads();
    }
    // ---- progress: wavefront step s of this workgroup is done (raised on idle steps too: counters only go up)
    if (tid == 0) __hip_atomic_store(f_mine, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (row_used) {
    const size_t hc = (size_t)g_row * H + g_unit;
    U.c[hc] = c_reg[0]; U.c[hc + 1] = c_reg[1];
    U.h_last[hc] = h_reg[0]; U.h_last[hc + 1] = h_reg[1];
  }
}

#ifndef LQ3_RING1
#define LQ3_RING1 6
#endif
#ifndef LQ3_RING2
#define LQ3_RING2 4
#endif

template <int RTS, int D>
static hipError_t launch_midseq_cfg(LstmMidSeqArgs& a, dim3 grid, hipStream_t stream, bool* fits) {
  const void* fn = reinterpret_cast<const void*>(lstm_midseq_x3_kernel<RTS, D>);
  int capacity = 0;
  if (hipError_t e = coresident_blocks(fn, lq3::NT, lq3::LDS_BYTES, &capacity)) return e;
  *fits = (int)(grid.x * grid.y * grid.z) <= capacity;
  if (!*fits) return hipSuccess;
  void* params[] = {&a};
  if (hipLaunchCooperativeKernel(fn, grid, dim3(lq3::NT), params, (unsigned)lq3::LDS_BYTES, stream) != hipSuccess) {
    (void)hipGetLastError();   // no cooperative launch in this context (stream capture): the caller steps launch by launch
    *fits = false;
  }
  return hipSuccess;
}

size_t lstm_midseq_flag_uints(int n_units, int H) { return (size_t)n_units * (H / lq3::BU); }

bool lstm_midseq_shape_ok(int B, int H, int n_units, const int* ks_in) {
  if (B < 1 || B > lq3::BM || H % 32 != 0 || n_units < 1 || n_units > 4) return false;
  for (int u = 0; u < n_units; ++u)
    if (ks_in[u] + H / 16 > 4 * lq3::MAXW) return false;
  return true;
}

// *done = false: not covered or not launchable here (the caller then steps the wavefront launch by launch).  `flags`
// (lstm_midseq_flag_uints) is zeroed here.
hipError_t launch_lstm_midseq_x3(const LstmMidSeqArgs& a_in, hipStream_t stream, bool* done) {
  *done = false;
  LstmMidSeqArgs a = a_in;
  int ks_in[4];
  for (int u = 0; u < a.n_units && u < 4; ++u) ks_in[u] = a.unit[u].ks_in;
  if (!lstm_midseq_shape_ok(a.B, a.H, a.n_units, ks_in)) return hipSuccess;
  a.spin_limit = options().spin_limit > 0 ? options().spin_limit : 1 << 20;
  a.timeouts = poll_timeout_word();
  if (!a.timeouts) return hipErrorOutOfMemory;
  if (hipError_t e = hipMemsetAsync(a.flags, 0, lstm_midseq_flag_uints(a.n_units, a.H) * sizeof(unsigned), stream)) return e;
  dim3 grid(a.H / lq3::BU, 1, a.n_units);
  if (a.B <= 32) return launch_midseq_cfg<1, LQ3_RING1>(a, grid, stream, done);
  return launch_midseq_cfg<2, LQ3_RING2>(a, grid, stream, done);
}

}  // namespace empose
