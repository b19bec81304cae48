// The LSTM wavefront step of large batches on three bf16 pieces per operand (bf16x3.h), second form (round 6; the first is
// lstm_x3.hip, whose operand formats -- weights in fragment order [k-step][32-unit block][gate][piece], activations as A planes
// [32-row tile][k-step][piece] written by the producing step -- this kernel shares unchanged; reference nn/layers.py:133-157).
//
// What lstm_chain_x3_kernel loses, by its own stamps and counters (profiles/r05f_lstm_x3_lab.txt, r05f_sq_counters_*):
//   * its four waves split K, so every wave streams the whole 64 x 128 operand block of its k-steps from L2: 18 KB per
//     48 MFMAs and wave, 47 bytes per clock and CU -- three quarters of what an XCD's L2 delivers; the products take 29 k
//     clocks where the matrix pipe needs 20 k;
//   * the K split ends in a 4 x 34 KB exchange of partial sums through LDS and a finish that re-reads them: 7.5 k clocks
//     per unit with no MFMA in flight, twice per launch.
// Here a workgroup owns 128 batch rows x 32 hidden units x 4 gates of ONE unit (layer, time step) of the launch and its
// four waves split the ROWS: wave w keeps rows 32 w .. 32 w + 31 x all 128 gate columns in 64 accumulator registers.
//   * The weight block of a k-step (12 fragments = 12 KB, contiguous in the packed matrix) is fetched ONCE per workgroup --
//     every thread moves three 16-byte pieces global -> registers -> LDS, two k-steps ahead of the LDS reads, four ahead of
//     the products -- and read back as 12 conflict-free ds_read_b128 per wave; a wave's own A fragments (3 per k-step) come
//     straight from the planes.  Per k-step and workgroup 24 KB leave L2 for 96 MFMAs (the K-split kernel: 72 KB).
//   * A wave holds all four gates of its (row, unit) cells in the same lane and accumulator index, so the cell update runs
//     in registers: no exchange, no barrier; c, h and y leave as 128-byte row segments, and the new hidden state's three
//     pieces leave in A-fragment order after a 4 KB transpose through the wave's own LDS.
// One raw s_barrier per TWO k-steps hands the LDS stage over (lgkmcnt only: the global loads in flight stay in flight).
// One workgroup per CU by its LDS request (one wave per SIMD: scripts/dev/bf16_hazard_repro.md).
// The two units of a launch (layer 0: K = input + H, layer 1: K = 2 H) go to different workgroups (blockIdx.z).
#include "bf16x3.h"
#include "gemm_epilogue.h"

namespace empose {

namespace lr {
constexpr int BM = 128, BU = 32, NT = 256;
constexpr int FRAG = 512;                                // bf16 elements of one wave fragment (1 KB)
constexpr int WBLK = 12 * FRAG;                          // elements of one k-step's weight block (4 gates x 3 pieces)
constexpr int WBUF_BYTES = WBLK * 2;                     // 12,288
constexpr int TP = 33;                                   // row stride of the transpose tile (floats)
constexpr int NBUF = 8;                                   // LDS stage buffers (a k-step's weight block each)
constexpr size_t LDS_BYTES = NBUF * WBUF_BYTES;           // 96 KB > half of a CU's 160 KB: never two workgroups on a CU
constexpr int SG_MFMA = 0x008, SG_VMEM_RD = 0x020, SG_DS_RD = 0x100, SG_DS_WR = 0x200;
static_assert(LDS_BYTES > 80 * 1024 && 4 * 32 * TP * 4 <= 2 * WBUF_BYTES, "LDS layout");
}  // namespace lr

typedef const __attribute__((address_space(1))) u32x4_t* lr_gvec_t;
typedef const __attribute__((address_space(1))) unsigned short* lr_gptr_t;

#define LR_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

__device__ __forceinline__ float lr_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float lr_tanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }

__device__ unsigned short lr_zero_frags[3 * lr::FRAG];   // zero-initialised: the A fragments of the padding k-steps

__global__ __launch_bounds__(lr::NT) void lstm_rows_x3_kernel(LstmX3Args a) {
  X3_EXCLUSIVE_SIMD();
  using namespace lr;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int H = a.H, B = a.B, F = a.F;
  const int JB = H / BU;
  int jb = blockIdx.x, rb = blockIdx.y, uz = blockIdx.z;
#ifndef LR_LAB_NOXCD
  // Workgroups go to the 8 XCDs round-robin by their linear index.  When the grid is 16 unit blocks x 8 row blocks (the
  // headline shape) the 32 workgroups of an XCD are remapped to 4 unit blocks x 4 row blocks x both units: its L2 then
  // streams 5.2 + 5.2 MB of weights and A planes per launch instead of 2.6 + 10.3 MB.
  if (JB == 16 && gridDim.y == 8) {
    const int lin = blockIdx.x + 16 * (blockIdx.y + 8 * blockIdx.z);
    const int xcd = lin & 7, s = lin >> 3;
    if ((int)gridDim.z == 2) { jb = 4 * (xcd & 3) + (s & 3); rb = 4 * (xcd >> 2) + ((s >> 2) & 3); uz = s >> 4; }
  }
#endif
  const int j0 = jb * BU;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const LstmX3Unit& U = a.unit[uz];
  const int RT = (B + 31) / 32;
  const int rt_own = rb * 4 + wave;
  const bool tile_live = rt_own < RT;                 // (a row tile past the batch: fetches the last one, stores nothing)
  const int rt = tile_live ? rt_own : RT - 1;
  const int KS_h = H / 16, KS_in = U.ks_in, KS = KS_in + KS_h;
  const int t = U.t;
  const int unit = j0 + l31;

  // ---- what the cell update reads besides the sums, fetched now (latency under the K loop): the old cell state of the
  // lane's 16 (row, unit) cells, the four biases of its unit, which of its rows are inside their sequence
  float c_old[16], e_bias[4];
  unsigned live_mask = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    const int rowc = row < B ? row : B - 1;
    c_old[r] = U.c[(size_t)rowc * H + unit];
    const int len = a.seq_lengths ? a.seq_lengths[rowc] : F;
    live_mask |= (t < len ? 1u : 0u) << r;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) e_bias[q] = U.bias[q * H + unit];

  f32x16 acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  // ---- operand addressing: k-step g of the unit's K = the input's steps, then the recurrent ones
  const unsigned short* const p_in = U.a3_in; const unsigned short* const p_rec = U.a3_rec;
  const unsigned short* const p_wih = U.w3_ih; const unsigned short* const p_whh = U.w3_hh;
  u32x4_t sreg[3][3];        // the thread's three 16-byte pieces of a weight block, on their way to LDS
  u32x4_t areg[3][3];        // the wave's A fragments (three pieces) of a k-step
  u32x4_t wreg[2][12];       // a k-step's weight fragments [gate][piece], read back from LDS

  auto gload_w = [&, p_wih, p_whh](u32x4_t (&S)[3], int g) {
    g = g < KS ? g : KS - 1;                          // (past the last step: fetched, never multiplied)
    const bool in = g < KS_in;
    const int ks = in ? g : g - KS_in;
    lr_gptr_t wb = (lr_gptr_t)(in ? p_wih : p_whh) + ((size_t)ks * JB + jb) * WBLK + tid * 8;
#pragma unroll
    for (int p = 0; p < 3; ++p) S[p] = *(lr_gvec_t)(wb + p * (NT * 8));
  };
  auto gload_a = [&, p_in, p_rec](u32x4_t (&A)[3], int g) {
    // The K loop runs whole groups of six steps (its register sets rotate with periods 3, 3 and 2; exits in mid-group cost
    // the compiler 170 spilled registers): the steps past the unit's last one multiply ZERO A fragments -- at most five
    // k-steps of 41 / 64 added, 1.6 / 3 % -- against the last step's weights again.
    // (they are FETCHED as zeros, from lr_zero_frags: masking loaded values made the compiler wait for the load a step early)
    const bool dead = g >= KS;
    g = dead ? KS - 1 : g;
    const bool in = g < KS_in;
    const int ks = in ? g : g - KS_in, ksn = in ? KS_in : KS_h;
    lr_gptr_t ab = dead ? (lr_gptr_t)lr_zero_frags + lane * 8
                        : (lr_gptr_t)(in ? p_in : p_rec) + (((size_t)rt * ksn + ks) * 3) * FRAG + lane * 8;
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) A[pc] = *(lr_gvec_t)(ab + pc * FRAG);
  };
  auto lds_write = [&](const u32x4_t (&S)[3], int g) {
    unsigned char* b = smem + (g & (NBUF - 1)) * WBUF_BYTES + tid * 16;
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4_t*>(b + p * (NT * 16)) = S[p];
  };
  auto lds_read = [&](u32x4_t (&W)[12], int g) {
    const unsigned char* b = smem + (g & (NBUF - 1)) * WBUF_BYTES + lane * 16;
#pragma unroll
    for (int f = 0; f < 12; ++f) W[f] = *reinterpret_cast<const u32x4_t*>(b + f * 1024);
  };
  auto mma = [&](const u32x4_t (&A)[3], const u32x4_t (&W)[12]) {
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, A[X3_PA[p]]),
                                                         __builtin_bit_cast(bf16x8_t, W[q * 3 + X3_PB[p]]), acc[q], 0, 0, 0);
  };
  auto handover = [&]() {
#ifndef LR_LAB_NOBARRIER   // (scripts/dev/lstm_rows_lab.sh: timing ablations, results are then wrong)
    __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): this wave's LDS writes and reads of the stage are done
    __builtin_amdgcn_s_barrier();
#endif
  };
  // One k-step i: the products of step i (operands in registers) with, spread between them, the global loads of the weight
  // pieces of step i + 6 and the A fragments of step i + 2, the LDS write of the pieces of step i + 4 (loaded two steps ago)
  // and the LDS reads of the fragments of step i + 1.  The stage is handed over every SECOND step: a block is written at
  // step k - 4 and read at step k - 1 (a hand-over lies between), its buffer is rewritten at step k + 4 (eight buffers).
  auto step = [&](int i, bool sync, u32x4_t (&S_new)[3], const u32x4_t (&S_ready)[3], u32x4_t (&A_new)[3],
                  const u32x4_t (&A_cur)[3], u32x4_t (&W_next)[12], const u32x4_t (&W_cur)[12]) {
#ifndef LR_LAB_NOGLOAD
    gload_w(S_new, i + 6);
    gload_a(A_new, i + 2);
#endif
#ifndef LR_LAB_NOLDS
    lds_write(S_ready, i + 4);
    lds_read(W_next, i + 1);
#endif
    mma(A_cur, W_cur);
#pragma unroll
    for (int q = 0; q < 24; ++q) {
      LR_SGB(SG_MFMA, 1);
      if (q < 6) LR_SGB(SG_VMEM_RD, 1);
      else if (q < 9) LR_SGB(SG_DS_WR, 1);
      else if (q < 21) LR_SGB(SG_DS_RD, 1);
    }
    if (sync) handover();
  };

  // ---- prologue: steps 0 .. 3 in LDS, 4 and 5 in flight, the fragments of step 0 in registers
  gload_w(sreg[0], 0);
  gload_w(sreg[1], 1);
  gload_w(sreg[2], 2);
  gload_a(areg[0], 0);
  gload_a(areg[1], 1);
  lds_write(sreg[0], 0);
  lds_write(sreg[1], 1);
  lds_write(sreg[2], 2);
  gload_w(sreg[0], 3);
  lds_write(sreg[0], 3);
  gload_w(sreg[1], 4);
  gload_w(sreg[2], 5);
  handover();
  lds_read(wreg[0], 0);

#if defined(LR_LAB_KS)      // (lab: a fixed number of k-steps: what a launch costs besides its K loop)
  const int KS_loop = LR_LAB_KS;
#elif defined(LR_LAB_BALANCED)   // (lab: every workgroup runs the mean of the two units' k-steps: the floor a balanced split would have)
  const int KS_loop = 54;
#else
  const int KS_loop = KS;
#endif
  // (sets rotate with periods 3, 3 and 2: six steps per trip)
  for (int i = 0; i < KS_loop; i += 6) {
    step(i, false, sreg[0], sreg[1], areg[2], areg[0], wreg[1], wreg[0]);
    step(i + 1, true, sreg[1], sreg[2], areg[0], areg[1], wreg[0], wreg[1]);
    step(i + 2, false, sreg[2], sreg[0], areg[1], areg[2], wreg[1], wreg[0]);
    step(i + 3, true, sreg[0], sreg[1], areg[2], areg[0], wreg[0], wreg[1]);
    step(i + 4, false, sreg[1], sreg[2], areg[0], areg[1], wreg[1], wreg[0]);
    step(i + 5, true, sreg[2], sreg[0], areg[1], areg[2], wreg[0], wreg[1]);
  }
  // (every trip ends in a hand-over: no wave still reads or writes the stage buffers, which now hold the transpose tiles)

  // ---- the cell update, in registers: lane = (unit, 16 rows), accumulator q = gate q (PyTorch's order i, f, g, o)
  float hv[16];
  const bool carry = a.seq_lengths != nullptr;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float g_i = lr_sigmoid(acc[0][r] + e_bias[0]), g_f = lr_sigmoid(acc[1][r] + e_bias[1]);
    const float g_g = lr_tanh(acc[2][r] + e_bias[2]), g_o = lr_sigmoid(acc[3][r] + e_bias[3]);
    const float c_new = g_f * c_old[r] + g_i * g_g;
    hv[r] = g_o * lr_tanh(c_new);
    c_old[r] = c_new;                                   // (rows past their length keep the old state: not stored below)
  }
  float* tile = reinterpret_cast<float*>(smem) + wave * (32 * TP);
  if (tile_live) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rl = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int row = rt * 32 + rl;
      const bool live = (live_mask >> r) & 1u;
      float h_state = hv[r];
      if (row < B) {
        const size_t hc = (size_t)row * H + unit;
        if (live) U.c[hc] = c_old[r];
        else h_state = carry ? U.h_prev[hc] : 0.f;      // past its length: the state is carried, the output is zero
        U.h_next[hc] = h_state;
        if (U.y) U.y[((size_t)row * F + t) * U.y_ld + U.y_col + unit] = live ? hv[r] : 0.f;
      }
      tile[rl * TP + l31] = h_state;
    }
    // the new hidden values as pieces, where the next step's (and the layer above's) A fragments expect them: lane =
    // (row, 8 consecutive units), one whole 1 KB fragment per store instruction and piece
    const int row = rt * 32 + l31;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const float* src = tile + l31 * TP + p * 16 + lh * 8;
      const Pieces q = split8(src[0], src[1], src[2], src[3], src[4], src[5], src[6], src[7]);
      if (row < B) {
        unsigned short* o = U.a3_out + (((size_t)rt * KS_h + (j0 >> 4) + p) * 3) * FRAG + lane * 8;
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) *reinterpret_cast<u32x4_t*>(o + pc * FRAG) = q.p[pc];
      }
    }
  }
}

bool lstm_rows_x3_covers(int H) { return H % 32 == 0; }

hipError_t launch_lstm_rows_x3(const LstmX3Args& a, hipStream_t stream) {
  if (a.n_units == 0) return hipSuccess;
  if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(lstm_rows_x3_kernel), lr::LDS_BYTES)) return e;
  dim3 grid(a.H / lr::BU, (a.B + lr::BM - 1) / lr::BM, a.n_units);
  hipLaunchKernelGGL(lstm_rows_x3_kernel, grid, dim3(lr::NT), lr::LDS_BYTES, stream, a);
  return hipGetLastError();
}

}  // namespace empose
