// Internal launch interfaces of the HIP kernels (gfx950 only). Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <initializer_list>
#include <utility>
#include <vector>

namespace empose {

// Kernel-variant selection for A/B and bit-identity tests (empose_set_option in the C ABI).  Plain process-wide ints set
// by an explicit call -- the library never reads the environment.  Defaults as initialised below (empose_reset_options
// restores them).
struct Options {
  int mlp_fused = 1;      // one-launch LDS-resident update MLPs (0: layer by layer)
  int lstm_persist = 1;   // whole-sequence small-batch LSTM kernel (0: step launches)
  int gemm_splitk = 1;    // split-K tile for problems of few output tiles (0: generic tiles)
  int heads_rows = 1;     // large batches: both init heads as one row-block product (0: two problems on the generic tile)
  int lstm_seq = 0;       // large batches: the whole sequence in one cooperative launch (measured slower: 0 = a launch per wavefront step)
  int bptt_wave = 1;      // training: the reverse recurrences of a 2-layer LSTM as a wavefront (0: layer after layer)
  int gemm_wide = 1;      // 256 x 256 four-wave tile (0: generic tiles)
  int smpl_tile = 1;      // frame-per-lane SMPL sub-mesh kernel: 0 never, 1 from 16384 frames on, 2 always
  int smpl_fuse = 1;      // on that path: update + feature row / Rodrigues reverse inside the blend GEMMs (0: own kernels)
  int train_fused = 0;    // train-mode BatchNorm / PReLU folded into the GEMMs: 0 never (default: measured no faster,
                          // DESIGN.md), 1 above 1024 rows, 2 always
  int atb_target = 0;     // workgroups the A^T B weight-gradient GEMM aims for when it splits its reduction (0: by size)
  int atb_fast = 1;       // A^T B: whole-tile / whole-chunk problems on the branch-free interior kernel (0: the general kernel)
  int atb_chunk = 0;      // rows per staged chunk of that kernel, 16 or 32 (0: by size)
  int mesh_skin_mfma = 0; // split-bf16 full-mesh variant: the bone blend as a second matrix-core contraction (mesh_rows_bf16s_kernel;
                          // measured 9 % SLOWER than vector skinning: 15.6 against 17.1 M frames/s, so off by default)
  int train_epi = 1;      // train-mode MLP layer: BatchNorm statistics in the GEMM epilogues + ONE combine-and-apply launch per
                          // layer and direction (train_fused.hip, finish kernels): 0 never, 1 above BN_SINGLE_PASS_ROWS rows, 2 always
  int spin_limit = 0;     // polls of the cooperative LSTM kernels give up after this many spins (0: their own limits)
  int rows_x3 = 1;        // the row-block products with fused prologue / epilogue (blend GEMMs of the frame-per-lane SMPL path, init
                          // heads) on the same three-piece bf16 arithmetic; 0: the fp32 MFMA instruction
  int lstm_x3 = 1;        // large-batch LSTM steps (inference, uni-directional): fp32 products as six bf16-MFMA products of three
                          // bf16 pieces per operand; 1: lstm_x3.hip (K-split waves); 2: lstm_rows_x3.hip (row-split waves, weights
                          // through LDS, cell update in registers -- round 6, measured 10 % SLOWER: 47.5 against 42.9 us per
                          // launch, profiles/r06_lstm_rows_lab.txt, so opt-in); 0: the fp32 MFMA instruction (lstm_chain_kernel)
  int train_cols = 1;     // training at <= 512 rows: a layer's product + BatchNorm + PReLU as one launch, both update networks
                          // side by side (train_cols.hip); 0: a product and a BatchNorm launch per layer and network
  int cols_coop = 1;      // those one-launch layers, eager: launched with hipLaunchCooperativeKernel (all workgroups resident by
                          // the runtime's guarantee); 0: the ordinary launch (residency argued from the occupancy query)
  int mesh_x3 = 1;        // full-mesh vertices: blend shapes as six bf16-MFMA products of three bf16 pieces per operand
                          // (mesh_x3.hip): 1 stores staggered into the next tile's products, 2 stores at the end of the tile,
                          // 3 skinning software-pipelined under the next tile's products; 0: the fp32 MFMA instruction
                          // (mesh_rows_kernel)
  int train_x3 = 1;       // training at more than 1024 rows: the layer products y = a W^T and dA = dY W of the update networks on
                          // three bf16 pieces (gemm_train_x3_kernel) when the caller supplies the packed weights
                          // (empose_mlp_params::weight_x3 / weight_t_x3); 0: the fp32 MFMA tile
  int lstm_midseq = 0;    // 1: medium batches (4 .. 64 rows, inference): the whole sequence in one cooperative launch with the
                          // weight pieces in registers (lstm_midseq_x3.hip) -- built, same bits, measured SLOWER than a launch
                          // per wavefront step (10.5 against 8.2 us per step at 32 rows: a hand-over between XCDs costs more
                          // than a kernel boundary; profiles/r06i_lstm_midseq_lab.txt), so opt-in
  int lstm_mid16 = 1;     // LSTM steps of 17 .. 64 rows: 64 rows x 4-unit tiles on all 256 CUs (lstm_mid16_x3.hip); 0: the 8-unit
                          // tiles of lstm_mid_x3.hip (128 workgroups)
  int lstm_mid_x3 = 1;    // LSTM steps of 17 .. 256 rows (inference, uni-directional) on three bf16 pieces, 64 x 8-unit tiles
                          // (lstm_mid_x3.hip); 0: lstm_mid_kernel (fp32 MFMA, operands through LDS)
  int lstm_fewrows = 1;   // LSTM steps of 4 .. 16 rows: all threads of a workgroup split K, lane reduce-scatter (lstm_fewrows_kernel),
                          // instead of the whole-sequence kernel / lstm_small_kernel (0: those; they share their bits)
  int mlp_x3 = 1;         // fused update MLPs: fp32 products as six bf16-MFMA products of three bf16 pieces per operand
                          // (mlp_fused_x3.hip: fp32-equivalent accuracy, measured equal to the fp32 instruction's against
                          // float64); 0: the fp32 MFMA instruction (mlp_fused.hip); 2: the variant whose waves share the
                          // A-side split through LDS (a barrier per k-step; measured 9 % slower)
};
Options& options();

// Dynamic LDS above the 64 KB a kernel may use by default: hipFuncAttributeMaxDynamicSharedMemorySize of `fn` on the
// CURRENT device, raised when `bytes` exceeds what was already allowed there (cached per (device, kernel): a process that
// drives several GPUs sets it on each).  More than a CU has (LDS_BYTES_PER_CU) is refused here instead of at the launch.
// The cooperative whole-sequence LSTM kernels poll exchange words written by other workgroups; a poll that exceeds its
// spin limit gives up, poisons the outputs with NaN (it does not hang) and counts itself in ONE host-mapped word that the
// entry points read without synchronising: the next empose_* call that runs a recurrence returns EMPOSE_ETIMEOUT, and
// empose_async_status() reports it to a caller that has synchronised.  Option "spin_limit" (0 = the kernels' own limits)
// forces a limit, for tests.
unsigned* poll_timeout_word();          // device-visible address of the counter
unsigned poll_timeouts_take();          // host: read and clear
unsigned poll_timeouts_peek();          // host: read
constexpr size_t LDS_BYTES_PER_CU = 160 * 1024;
hipError_t allow_dynamic_lds(const void* fn, size_t bytes);
// Workgroups of `fn` (block size `threads`, `lds_bytes` of dynamic LDS) that can be resident at once on the CURRENT device:
// what a cooperative launch may ask for.  Cached per (device, kernel); the attribute above is set on the way.
hipError_t coresident_blocks(const void* fn, int threads, size_t lds_bytes, int* blocks);

// ---------------------------------------------------------------------------------------------------------------
// fp32 matrix-core linear layer:  C[m][n] = act( (sum_k A[m][k] * W[n][k]) * scale[n] + shift[n] ) (+ resid[m][n])
// ---------------------------------------------------------------------------------------------------------------
struct GemmProb {
  const float* A; int lda;
  const float* W; int ldw;
  float* C; int ldc;
  int M, N, K;              // K % 4 == 0, lda/ldw % 4 == 0, A/W 16-byte aligned
  const float* scale;       // [N] or nullptr (=1)
  const float* shift;       // [N] or nullptr (=0)
  const float* resid; int ldr;  // added after the activation (skip connections) or nullptr
  int act;                  // 0 none, 1 PReLU(slope) then + resid, 2 + resid then ReLU
  float slope;
  // Train-mode layer fusion (gemm_tn_f32_kernel<Cfg, 2> only; see TrainGemmArgs / train_fused.hip):
  const float* a_s = nullptr; const float* a_t = nullptr; const float* a_slope = nullptr;   // A' = PReLU(s[k] A + t[k])
  float* stat_part = nullptr;                                                               // column statistics of C
  const float* e_y = nullptr; int ld_ey = 0;                                                // product is dA: C = dyh, sums
  const float* e_s = nullptr; const float* e_t = nullptr; const float* e_mean = nullptr; const float* e_rstd = nullptr;
  const float* e_slope = nullptr;
};
struct GemmBatch { GemmProb p[2]; int count; int role = 0; int xcd_swizzle = 1; };  // role 1 = update-net hidden layer (profiling name only)

// The same product with K split over workgroups (a few hundred rows, long K, small output: the recurrent products of
// back-propagation through time); `workspace`: gemm_ksplit_workspace_floats floats.  Deterministic (partials added in
// slice order by a second kernel).
bool gemm_ksplit_applicable(int M, int N, int K);
size_t gemm_ksplit_workspace_floats(int M, int N, int K);
struct LstmCellBwdArgs;
// `cell` (back-propagation through time): the product is dh of the step below (N = H, one value per (row, unit)); the
// reduce kernel runs that step's cell on it instead of storing C.
hipError_t launch_gemm_ksplit(const GemmProb& p, float* workspace, hipStream_t stream, const LstmCellBwdArgs* cell = nullptr);
// At most 16 rows: the matrix-vector kernel, same epilogue.
bool gemm_fewrows_applicable(int M, int N, int K);
hipError_t launch_gemm_fewrows_cell(const GemmProb& p, const LstmCellBwdArgs& cell, hipStream_t stream);

// Back-propagation through time as a wavefront over the layers: one launch forms the incoming hidden-state cotangent
// of up to two (layer, step) units and runs their cells.  A unit's product may have two K segments with their own
// operands -- dh0_t = dG0_{t+1} . W_hh0 + dG1_t . W_ih1 is one K = 8H product over the rows [dG0_{t+1} | dG1_t] that
// are never concatenated in memory.  C = sum_seg A_seg . W_seg^T (+ resid); the unit's cell consumes it.
struct RecSeg { const float* A; int lda; const float* W; int ldw; int K; };   // A [M][lda], W [N][ldw], K % 4 == 0
struct RecProb { RecSeg seg[2]; int nseg; int M, N; const float* resid; int ldr; };
struct RecBatch { RecProb p[2]; int count; };
size_t rec_ksplit_workspace_floats(int M, int N, int K_total_max, int count);
// K-split tiles (a few hundred rows; every segment's K a multiple of 256) / matrix-vector kernel (<= 16 rows, every
// segment's K in [1024, 2048]); cells[i] is the cell of problem i.
hipError_t launch_rec_ksplit(const RecBatch& b, const LstmCellBwdArgs* cells, float* workspace, hipStream_t stream);
hipError_t launch_rec_fewrows(const RecBatch& b, const LstmCellBwdArgs* cells, hipStream_t stream);

// C[M][N] = A . W^T (+ bias) with strided operands: A(m, k) = A[m * a_rs + k * a_ks], W(n, k) = W[n * w_rs + k * w_ks]
// (small problems only, split-K tile; `strided_gemm_applicable`).
struct StridedGemm {
  const float* A; long a_rs, a_ks;
  const float* W; long w_rs, w_ks;
  float* C; int ldc;
  const float* bias;
  int M, N, K;
};
bool strided_gemm_applicable(int M, int N);
hipError_t launch_strided_gemm(const StridedGemm& p, hipStream_t stream);

// Train-mode BatchNorm1d + PReLU, forward or backward (bn_prelu.hip).
struct BnPreluArgs {
  int M, C;
  const float* x; int ldx;
  const float* gamma; const float* beta; const float* slope;   // slope: one device float
  float eps, momentum;
  float* running_mean; float* running_var; long long* num_batches_tracked;   // forward only, may be null
  float* z; int ldz;                                                         // forward output
  float* save_mean; float* save_rstd;                                        // written by forward, read by backward
  const float* dz; int lddz; float* dx; int lddx;                            // backward
  float* dgamma; float* dbeta; float* dslope_partial;                        // [C], [C], [ceil(C / 32)] scratch
  float* dslope; int* counter;   // slope gradient (one float); arrival counter, zero before the first launch, self re-arming
  float* workspace = nullptr;    // bn_prelu_workspace_floats(M, C) floats: partial sums of the row-split path (large M)
  int accumulate = 0;            // backward: add to dgamma / dbeta / dslope instead of overwriting them
};
// One train-mode layer (product + BatchNorm + PReLU, forward or backward) of one or two MLPs as ONE launch at small
// batches (train_cols.hip): a workgroup per 16 columns x quarter of the rows, the column statistics exchanged between
// the row parts through a mailbox of tagged words.
struct ColsNet {
  const float* A; int lda;           // operand rows [M][K]: the layer input (forward) / the cotangent dZ_l (backward)
  const float* W; int ldw;           // [N][K]: W_l (forward) / W_l^T (backward: rows = columns of the layer below);
  int w_kmajor, Kw;                  // backward: W is W_l itself, [Kw][N] with Kw <= K rows -- no transposed copy needed
  const float* bias;                 // forward: [N]
  int N, K;                          // K % 4 == 0, lda % 4 == 0, ldw % 4 == 0
  const float* gamma; const float* beta; const float* slope;       // BatchNorm / PReLU of the N columns
  float* running_mean; float* running_var; long long* num_batches; // forward, may be null
  const float* z_in; float* z; int ldz;   // pre-BatchNorm rows: written by the forward, read (z_in) by the backward
  float* out; int ld_out;            // forward: the activations (mode 1: the plain product); backward: dZ of the N columns
  float* mean; float* rstd;          // [N]: written by the forward, read by the backward
  float* dgamma; float* dbeta; float* dslope;   // backward
};
struct ColsArgs {
  ColsNet net[2];
  int n_nets, M;
  float eps, momentum;
  int accumulate;
  unsigned tag;                      // unique among the launches that share the mailbox since it was zeroed; > 0
  unsigned long long* mailbox;       // cols_mailbox_words(widest N) words
  int R, tiles_per_part, s_max, slices_per_group, plain, spin_limit; unsigned* timeouts;   // set by launch_cols
};
constexpr int COLS_MAX_ROWS = 512;
size_t cols_mailbox_words(int n_max);
bool cols_launchable(int n_max, int n_nets);   // every workgroup of such a launch resident at once on this device
hipError_t launch_cols(ColsArgs a, int mode, hipStream_t stream);   // mode 0 forward, 1 forward without BatchNorm, 2 backward

constexpr int BN_SINGLE_PASS_ROWS = 1024;   // up to here one workgroup per 32 columns walks all rows
size_t bn_prelu_workspace_floats(int M, int C);
hipError_t launch_bn_prelu(const BnPreluArgs& a, bool backward, hipStream_t stream);

// ---------------------------------------------------------------------------------------------------------------
// Train-mode MLP layer with the BatchNorm / PReLU passes folded into the GEMMs (train_fused.hip)
// ---------------------------------------------------------------------------------------------------------------
struct TrainGemmArgs {       // C[M][N] = A'[M][K] . W[N][K]^T (+ bias), 64 x 128 tile
  const float* A; int lda;
  const float* W; int ldw;
  float* C; int ldc;
  int M, N, K;               // K % 4 == 0, lda / ldw / ld_ay % 4 == 0, 16-byte aligned operands
  const float* bias;         // [N] or nullptr
  // A operand as it is staged.  amode 1: A' = PReLU(s[k] A + t[k]) (a_slope: one device float)
  const float* a_s; const float* a_t; const float* a_slope;
  const float* a_y; int ld_ay; const float* a_c;   // (amode 2, dev: A' = c[k] A + c[K + k] Y[m][k] + c[2K + k])
  // epilogue.  emode 1: also per 32-row block and column the sum and centred sum of squares of C -> part [M/32][2][N];
  // emode 2: the product is dA: C = dyh = dA * PReLU'(e_s y + e_t) and part [M/32][3][N] = sums of dyh,
  // dyh * (y - mean) rstd, (yhat <= 0) dA yhat
  float* part;
  const float* e_y; int ld_ey; const float* e_s; const float* e_t; const float* e_mean; const float* e_rstd;
  const float* e_slope;
};
hipError_t launch_gemm_train(const TrainGemmArgs& p, int amode, int emode, hipStream_t stream);
// The same products (amode 0) on three bf16 pieces per operand for large batches (gemm_f32.hip, gemm_train_x3_kernel):
// `wfrag` = the weight matrix W [N][K] as pieces in fragment order, made by launch_pack_x3 once per optimiser step.
bool gemm_train_x3_applicable(int M, int N, int K);
hipError_t launch_gemm_train_x3(const TrainGemmArgs& p, const unsigned short* wfrag, int emode, hipStream_t stream);
size_t pack_x3_elems(int N, int K);
hipError_t launch_pack_x3(const float* W, int ldw, int N, int K, unsigned short* out, hipStream_t stream);
struct BnFusedFwdArgs {
  int M, C; const float* part;                        // [ceil(M / 32)][2][C]
  const float* gamma; const float* beta; float eps, momentum;
  float* running_mean; float* running_var; long long* num_batches_tracked;   // may be null
  float* mean; float* rstd; float* s; float* t;       // [C] each: statistics and the fused transform a = PReLU(s y + t)
};
hipError_t launch_bn_fused_combine_fwd(const BnFusedFwdArgs& a, hipStream_t stream);
struct BnFusedBwdArgs {
  int M, C; const float* part;                        // [ceil(M / 32)][3][C]
  const float* gamma; const float* mean; const float* rstd;
  float* dgamma; float* dbeta; float* dslope; float* dslope_partial;   // [C], [C], [1], [ceil(C / 64)] scratch
  float* coef;                                        // [3][C]: dY = coef0 dyh + coef1 y + coef2
  int accumulate;
};
hipError_t launch_bn_fused_combine_bwd(const BnFusedBwdArgs& a, hipStream_t stream);
size_t bn_fused_partial_floats(int M, int C);
// Round 4: combine + apply in one launch on materialised activations / cotangents (train_fused.hip, "finish" kernels).
struct BnFinishFwdArgs {
  int M, C; const float* part;                        // [ceil(M / 32)][2][C] from the GEMM's statistics epilogue
  const float* gamma; const float* beta; float eps, momentum;
  float* running_mean; float* running_var; long long* num_batches_tracked;   // may be null
  float* mean; float* rstd; float* s; float* t;       // [C] each (out)
  const float* y; int ldy;                            // the GEMM's output
  float* act; int ld_act;                             // a = PReLU(s y + t)
  const float* slope;
  int rows_per_block = 0;                             // (set by the launcher)
};
hipError_t launch_bn_finish_fwd(BnFinishFwdArgs a, hipStream_t stream);
struct BnFinishBwdArgs {
  int M, C; const float* part;                        // [ceil(M / 32)][3][C] from the dX GEMM's epilogue
  const float* gamma; const float* mean; const float* rstd;
  float* dgamma; float* dbeta; float* dslope;         // [C], [C], [1]
  float* dslope_partial; int* counter;                // [ceil(C / 32)] scratch; a zeroed int (re-arms itself)
  int accumulate;
  float* dyh; int ld;                                 // in: dA * PReLU'(yhat); out: dY (in place)
  const float* y; int ldy;
  int rows_per_block = 0;
};
hipError_t launch_bn_finish_bwd(BnFinishBwdArgs a, hipStream_t stream);
hipError_t launch_bn_fused_apply_bwd(float* dyh, const float* y, const float* coef, int M, int C, hipStream_t stream);

// Launches one grid covering all problems of the batch (blockIdx.y selects the problem).
hipError_t launch_gemm(const GemmBatch& batch, hipStream_t stream);
const char* gemm_kernel_name(int M, int N, int K, int count, int role);

// One or two whole MLPs (all layers) in one launch: a workgroup keeps the activations of 64 rows in LDS from the first
// layer to the last (mlp_fused.hip).
constexpr int FUSED_MAX_LAYERS = 8;
constexpr int FUSED_MAX_WIDTH = 512;     // widest layer output the four 128-column waves cover
struct FusedLayer {
  const float* W; int K, N;              // weights in fragment order (api.hip pack_fragments), K % 4 == 0, N <= FUSED_MAX_WIDTH
  const float* scale; const float* shift;
  float slope; int act;                  // 0 none, 1 PReLU
};
struct FusedNet {
  const float* x; int ldx;
  float* out; int ld_out;
  int n_layers;                          // the last layer writes `out`, the others stay in LDS
  FusedLayer layer[FUSED_MAX_LAYERS];
};
struct FusedMlpArgs { FusedNet net[2]; int count; int M; };
hipError_t launch_mlp_fused(const FusedMlpArgs& args, hipStream_t stream);
// The same launch with FusedLayer::W = three bf16 pieces per weight in bf16-MFMA fragment order (api.hip
// pack_fragments_x3_raw): mlp_fused_x3.hip.  Hidden widths must be multiples of 64.
hipError_t launch_mlp_fused_x3(const FusedMlpArgs& args, hipStream_t stream);
// One linear layer C = A . W^T with A's row block resident in LDS and W (fragment order) streamed from L2.
bool gemm_rows_applicable(int M, int N, int K);
hipError_t launch_gemm_rows(const float* A, int lda, const float* Wp, float* C, int ldc, int M, int N, int K,
                            hipStream_t stream);

// ---------------------------------------------------------------------------------------------------------------
// LSTM
// ---------------------------------------------------------------------------------------------------------------
struct LstmUnitArgs {        // one (layer, direction), selected by blockIdx.z
  const float* w_ih;   // [4H][in_k]
  const float* w_hh;   // [4H][H]
  const float* bias;   // [4H] = b_ih + b_hh
  int in_k;            // width of the unit's input
  const float* in_seq; int in_ld;  // stored input sequence [B][F][in_ld] (used when in_from < 0)
  int in_from;         // >= 0: the input is the hidden state of unit `in_from` after its step (stacked wavefront)
  int t_offset;        // the unit processes step k = s - t_offset
  int reverse;         // 1: at step k row b visits time len_b - 1 - k
  float* h[2];         // [B][H] ping-pong: step k reads h[k & 1], writes h[(k + 1) & 1]
  float* c;            // [B][H] updated in place
  float* y; int y_ld; int y_col;   // output sequence [B][F][y_ld], columns [y_col, y_col + H), or nullptr
  // Training forward (forward units only): what back-propagation through time needs, batch-major like y.
  float* sv_gates = nullptr;       // [B][F][4H] activated gates (i | f | g | o) of every step, or nullptr
  float* sv_c = nullptr;           // [B][F][H] cell state after the step (unchanged past a row's length)
  float* sv_hprev = nullptr;       // [B][F][H] hidden state BEFORE the step; the caller fills slot t = 0 (h_0)
};
struct LstmSeg {             // one operand segment (input part / recurrent part) of a unit's step; built on the host
  const float* a;            // A rows: a + row * lda (+ per-row time offset, below)
  const float* w;            // W rows: w + (gate * H + unit) * ldw
  int lda, ldw;
  int K;
  int tstride;               // != 0: reverse unit on a stored sequence; row b adds max(len_b - 1 - k, 0) * tstride
  int k;                     // the unit's step index
  int ntiles;                // ceil(K / 64)
  int unit;                  // index into LstmWaveArgs::unit
  int pad;
};
struct LstmWaveArgs {
  LstmUnitArgs unit[4];
  int n_units;
  const int* seq_lengths;    // [B] or nullptr (required for reverse units)
  int B, F, H;
  int s;                     // launch index
  // set by launch_lstm_wave: blockIdx.z walks through the segments [z_beg[z], z_beg[z] + z_cnt[z]) (two per active unit)
  LstmSeg seg[8];
  int z_beg[4], z_cnt[4];
};
hipError_t launch_lstm_wave(const LstmWaveArgs& a, hipStream_t stream);
// Whole sequence of a stacked uni-directional LSTM on a LARGE batch (B > 256) in one cooperative launch (lstm.hip,
// lstm_seq_kernel): the units' hidden states rotate through three buffers (h[0], h[1] of the unit + a third one), the
// workgroups of a 64-row group synchronise through `counters` (lstm_seq_counter_uints(B) unsigned, zeroed by the launcher).
struct LstmSeqArgs {
  LstmUnitArgs unit[4];
  float* hs[4][3];
  int n_units;
  const int* seq_lengths;
  int B, F, H;
  unsigned* counters;
  int spin_limit;
  unsigned* timeouts;   // poll_timeout_word(): counts the polls that gave up (the outputs are NaN from there on)
};
// The wavefront step of large batches on the bf16 matrix path, three bf16 pieces per fp32 operand (lstm_x3.hip).  All
// operands arrive in fragment order: weights packed at model creation (api.hip pack_lstm_x3: [k-step][32-unit block][gate]
// [piece][512 bf16]), activations as A planes [32-row tile][k-step][piece][512 bf16] written by the producing step (hidden
// states) or by launch_lstm_split_rows (stored input, initial state).
struct LstmX3Unit {
  const unsigned short* w3_ih; const unsigned short* w3_hh;
  const float* bias;                 // [4H] = b_ih + b_hh
  const unsigned short* a3_in;       // the unit's input at this step: planes of x_t, or of the unit below's new hidden state
  int ks_in;                         // k-steps of 16 of that input
  const unsigned short* a3_rec;      // planes of h_{t-1}
  unsigned short* a3_out;            // planes of h_t
  const float* h_prev; float* h_next; float* c;   // [B][H] fp32: state hand-over and rows past their length
  float* y; int y_ld, y_col;         // output sequence [B][F][y_ld] or nullptr
  int t;                             // the unit's time step in this launch
};
struct LstmX3Args {
  LstmX3Unit unit[4];
  int n_units, units_per_block;      // blockIdx.z walks units [z * units_per_block, ...)
  const int* seq_lengths;
  int B, F, H;
};
hipError_t launch_lstm_chain_x3(const LstmX3Args& a, hipStream_t stream);
// the same step with the waves splitting ROWS, the weight block of a k-step shared through LDS and the cell update in
// registers (lstm_rows_x3.hip; option lstm_x3 = 2): one unit per workgroup, `units_per_block` unused
hipError_t launch_lstm_rows_x3(const LstmX3Args& a, hipStream_t stream);
// the step of MEDIUM batches (17 .. 256 rows): 64 rows x 8 units per workgroup, weights in the 8-unit-block order
// (api.hip pack_lstm_x3 with mid = true), K split over the waves (lstm_mid_x3.hip; option lstm_mid_x3)
hipError_t launch_lstm_mid_x3(const LstmX3Args& a, hipStream_t stream);
// ... at most 64 rows: 4-unit tiles (16 columns, v_mfma_f32_16x16x32_bf16), weights in the order of api.hip
// pack_lstm_x3_mid16, twice the workgroups (lstm_mid16_x3.hip; option lstm_mid16)
bool lstm_mid16_shape_ok(int B, int H);
hipError_t launch_lstm_mid16_x3(const LstmX3Args& a, hipStream_t stream);
// The whole sequence of a medium batch (B <= 64) in one cooperative launch with the weight pieces in registers
// (lstm_midseq_x3.hip; option lstm_midseq): same tiles, products and bits as the step launches of lstm_mid_x3.hip.
struct LstmMidSeqUnit {
  const unsigned short* w3_ih; const unsigned short* w3_hh;   // the 8-unit-block order of lstm_mid_x3.hip
  const float* bias;                 // [4H] = b_ih + b_hh
  const unsigned short* in3; size_t in_t_stride;   // layer 0: planes of the stored input, bf16 elements per time step
  int ks_in;                         // k-steps of 16 of the unit's input
  unsigned short* xa;                // [F + 1] sets of planes of the unit's hidden state: slot 0 = initial, slot t + 1 = h_t
  const float* h0; float* h_last; float* c;   // [B][H] fp32: initial hidden state, final hidden state, cell state (in place)
  float* y; int y_ld, y_col;         // output sequence [B][F][y_ld] or nullptr
};
struct LstmMidSeqArgs {
  LstmMidSeqUnit unit[4];
  int n_units;
  const int* seq_lengths;
  int B, F, H;
  unsigned* flags;                   // [n_units][H / 8] progress counters (lstm_midseq_flag_uints), zeroed by the launcher
  int spin_limit;
  unsigned* timeouts;                // poll_timeout_word()
};
size_t lstm_midseq_flag_uints(int n_units, int H);
bool lstm_midseq_shape_ok(int B, int H, int n_units, const int* ks_in);
hipError_t launch_lstm_midseq_x3(const LstmMidSeqArgs& a, hipStream_t stream, bool* done);
hipError_t launch_lstm_split_rows(const float* src, long row_stride, long z_stride, int n_z, int B, int K, int KS,
                                  unsigned short* dst, long dst_z_stride, hipStream_t stream);
constexpr int LSTM_SEQ_MIN_B = 257;   // below: lstm_mid_kernel / the small-batch kernels
size_t lstm_seq_counter_uints(int B);
hipError_t launch_lstm_seq(const LstmWaveArgs& a, float* const* h_third, unsigned* counters, hipStream_t stream, bool* done);
// Whole sequence of a stacked uni-directional LSTM in one cooperative launch (B <= 16); *done = false: not covered.
constexpr int LSTM_PERSIST_B = 16;   // largest batch of the whole-sequence kernel
constexpr int LSTM_FEWROWS_MIN_B = 4;   // from here to 16 rows a step launch of lstm_fewrows_kernel beats it (option lstm_fewrows)
size_t lstm_persist_xch_floats(int n_units, int B, int H);   // its exchange buffer (8-byte aligned)
hipError_t launch_lstm_persist(const LstmWaveArgs& a, float* xch, hipStream_t stream, bool* done);

// ---------------------------------------------------------------------------------------------------------------
// Training backward (train.hip)
// ---------------------------------------------------------------------------------------------------------------
constexpr int ATB_MAX_SEG = 8;
struct AtbArgs {             // C[N][ldc] = A[M][lda]^T . B[M][ldb]  (+ bias[n] = sum_m A[m][n])
  const float* A; int lda;
  const float* B; int ldb;
  float* C; int ldc;
  float* bias;               // [N] or nullptr
  int M, N, K;
  int accumulate;            // 1: C += A^T B, bias += column sums (gradient accumulation over the LGD iterations)
  // Row segments: with n_seg > 0 the M rows are n_seg blocks of seg_rows rows (a multiple of 32) that live at
  // A_seg[s] / B_seg[s] (same leading dimensions) -- the operands of one layer over the applications of the LGD loop.
  int n_seg = 0, seg_rows = 0;
  const float* A_seg[ATB_MAX_SEG] = {};
  const float* B_seg[ATB_MAX_SEG] = {};
  // Operand transform of the fused train-mode layer (train_fused.hip), per segment (index 0 without segments):
  //   b_mode 1: B' = PReLU(Bs[k] B + Bs[K + k])   (the layer input a_{l-1} re-formed from y_{l-1}; b_slope: device float)
  int b_mode = 0;
  const float* Bs_seg[ATB_MAX_SEG] = {};
  const float* b_slope = nullptr;
  // set by launch_gemm_atb
  int S; float* partial; float* bias_partial;
};
hipError_t launch_window_mean(const float* in, int ld_in, float* out, int ld_out, int T, int F, int C, hipStream_t stream);
hipError_t launch_axpby2d(int rows, int cols, float alpha, const float* x, int ldx, float beta, const float* y, int ldy,
                          float* out, int ldo, hipStream_t stream);
hipError_t launch_lgd_assemble(int T, int d_in, const float* x0, int ld0, const float* pose, const float* shape, float* X,
                               int ldx, hipStream_t stream);
hipError_t launch_lgd_update(int B, int F, float step, int shape_avg, const float* pose, const float* d_pose,
                             const float* shape, const float* d_shape, float* pose_next, float* shape_next,
                             hipStream_t stream);
hipError_t launch_lgd_cotangent(int B, int F, int first, const float* d_pose, const float* d_shape, const float* vp,
                                const float* vs, const float* g_theta, int ld_g, const float* g_beta, int ld_gb, float* Dp,
                                float* Ds, float step, int shape_avg, float* dpad, float* dspad, hipStream_t stream);
struct LossArgs {
  int B, F, N1, n_markers;
  int used_slot[12];                       // column slot of virtual sensor m in the network input, or -1
  const float* pose_hist; const float* shape_hist; const float* pos_hist; const float* ori_hist;   // [N1][T][66|10|36|108]
  const float* joints_final;               // [T][66]
  const float* pose_gt; const float* shape_gt; const float* joints_gt;   // [T][66], [B][10], [T][66] or nullptr
  const float* x_in; int ldx;              // measured sensors in the network input layout
  const int* seq_lengths; const float* masks;   // [B] or nullptr, [T][12] or nullptr
  float w_pose, w_shape, w_fk, w_rec;
  float* d_pose; float* d_shape; float* d_pos; float* d_ori; float* d_joints;
  float* partial;                          // [4][N1 * T]
  float* loss_vals;                        // [5]
};
hipError_t launch_lgd_losses(const LossArgs& a, hipStream_t stream);
constexpr int ADAM_CHUNK = 4096;
struct AdamArgs {
  void* const* params; void* const* grads; void* const* exp_avg; void* const* exp_avg_sq;   // device pointer tables
  const long long* sizes; const int* chunk_tensor; const long long* chunk_offset;
  float beta1, beta2, eps, step_size, inv_sqrt_bc2;
};
hipError_t launch_adam(const AdamArgs& a, int n_chunks, hipStream_t stream);
int atb_splits(int M, int N, int K);
size_t atb_workspace_floats(int M, int N, int K);
size_t atb_workspace_floats_max(int M, std::initializer_list<std::pair<int, int>> products);   // over (N, K) pairs
hipError_t launch_gemm_atb(AtbArgs a, float* workspace, size_t workspace_floats, hipStream_t stream);
hipError_t launch_transpose(const float* src, int ld_src, float* dst, int ld_dst, int rows, int cols, hipStream_t stream);
hipError_t launch_add2(const float* a, const float* b, float* out, int n, hipStream_t stream);
struct LstmCellBwdArgs {
  const float* gates;        // [B][F][4H] saved activations
  const float* c_all;        // [B][F][H]
  const float* c0;           // [B][H] or nullptr (zeros)
  const float* dy; int ld_dy;   // [B][F][ld_dy] cotangent of the layer's output sequence, or nullptr
  const float* dh_in;        // [B][H] from step t + 1, or nullptr (zeros)
  float* dc;                 // [B][H] in: from step t + 1, out: for step t - 1
  float* dgates;             // [B][F][4H] pre-activation gradients (this step's rows are written)
  float* dh_carry;           // [B][H] out: dh_in of rows that had ended at this step, else 0
  const int* seq_lengths;
  int B, F, H, t;
};
hipError_t launch_lstm_cell_bwd(const LstmCellBwdArgs& a, hipStream_t stream);

// ---------------------------------------------------------------------------------------------------------------
// SMPL sub-mesh kernels
// ---------------------------------------------------------------------------------------------------------------
constexpr int CHAIN_CHUNK = 8;  // pairs per partial-sum chunk of the per-bone (vertex, weight) lists
constexpr int NB = 22;           // body joints

// LDS record of one frame in chain_sensors_kernel (shared with the host, which packs LDS offsets into the tables).
struct ChainLds {  // per-frame float offsets
  int rot, out, g, at, v, dv, u, total;
  // inside u (time-shared): fn | fg | scr   then   part (later: pd, fp64 prefix sums) | m | x
  int fn, fg, scr, part, pd, m, x;
};

__host__ __device__ inline ChainLds chain_layout(int nv, int ncp, int max_deg, int n_chunks) {
  // Live ranges (phases of chain_sensors_kernel): rot P0-P2 | out P0-P5 | g, at P2-P7 | v P3-P4c | dv P4d-P5 |
  // fn P4a-P4b, fg P4c-P4d, scr P4b-P4d | part P5-P5c | m P5c-P6a | pd P6a-P6b | x P6b-P7.  Records that are never
  // live together share their space (rot with the scratch area, dv with v, x with m): 6.9 KB per frame instead of
  // 9.7 KB, i.e. one more workgroup per CU.
  ChainLds l;
  int o = 0;
  l.out = o; o += ncp;
  l.g = o; o += NB * 12;    // per joint: G^R (9, row-major) | G^t (3)
  l.at = o; o += NB * 3;    // A^t = G^t - G^R J
  l.v = o; o += nv * 3;
  l.dv = l.v;
  o = (o + 1) & ~1;         // 8-byte alignment for the fp64 prefix sums (below)
  l.u = o;
  l.rot = l.u;
  const int nf = 12 * max_deg;
  l.fn = l.u; l.fg = l.fn + nf * 3; l.scr = l.fg + nf * 6;
  const int sz1 = nf * 9 + 12 * 9;
  const int pd_floats = (NB + 1) * 12 * 2;             // [23][12] doubles, aliases the chunk partial sums
  const int part_size = n_chunks * 12 > pd_floats ? n_chunks * 12 : pd_floats;
  l.part = l.u; l.pd = l.u; l.m = l.part + part_size; l.x = l.m;
  const int sz2 = part_size + NB * 12;
  int sz = sz1 > sz2 ? sz1 : sz2;
  sz = sz > NB * 9 ? sz : NB * 9;
  o += sz;
  l.total = (o + 3) & ~3;
  return l;
}


// Word offsets into the packed table blob that chain_sensors_kernel stages into LDS (all entries are 32-bit).
struct ChainTabs {
  int path_mask, sub_mask, parents;       // [22] each: ancestors-or-self mask, subtree mask, parent index
  int dfs_pos, sub_size;                  // [22] each: position of a joint in depth-first pre-order, size of its subtree
  int skin_idx, skin_w;                   // [nv*kb]
  int chunk_bone, chunk_beg;              // [n_chunks]: CHAIN_CHUNK-pair slices of the per-bone (vertex, weight) lists
  int bone_chunk_ptr;                     // [23]
  int bone_vert, bone_w;                  // [nnz]
  int s_center, s_helper, s_deg, s_faces; // [12], [12], [12], [12*max_deg*3]
  int inc_ptr, inc_code;                  // [nv+1], [n_inc]: per vertex, what contributes to its cotangent
  int total;
};

struct SmplTables {  // device pointers
  int n_sensors, nv, j_off, ncp, kb, max_deg;
  int n_chunks;
  const uint32_t* blob;
  ChainTabs off;

  const float* wc; const float* wct;
  const int* parents;
  const int* skin_idx; const float* skin_w;
  const int* bone_ptr; const int* bone_vert; const float* bone_w;
  const int* s_center; const int* s_helper; const int* s_deg; const int* s_faces;
  const int* path_ptr; const int* path; const int* sub_ptr; const int* sub;
};

// Pack the network input columns and the per-frame loss weight.
struct PackArgs {
  const float* marker_pos; const float* marker_oris;  // [T][36], [T][108]
  const float* marker_masks;                           // [T][12] or nullptr
  const int* seq_lengths;                              // [B] or nullptr
  float* x; int ldx;                                   // cols [0, 12*n_markers)
  float* frame_scale;                                  // [T]
  int B, F, n_markers;
  int marker_idx[12];
  int rows_as_unpadded = 0;   // 1: a ragged row counts as an unpadded window of its own length
  int suppress_missing = 0;   // 1: readings of sensors whose mask is not 1 are replaced by mask_value (the
  float mask_value = 0.f;     //    arithmetic of reference data/data.py:284-302: x * valid + mask_value * !valid)
};
hipError_t launch_pack_inputs(const PackArgs& a, hipStream_t stream);

// theta/beta update (+ window mean of the shape), Rodrigues and the GEMM feature row.
struct FeatArgs {
  float* theta; int ld_theta;          // [T][>=66] updated in place
  float* beta; int ld_beta;            // [T][>=10] updated in place
  const float* d_theta;                // [T][66] or nullptr
  const float* d_beta;                 // [T][10] or nullptr
  float theta_step;                    // theta += theta_step * d_theta
  float beta_keep, beta_step;          // beta = beta_keep * beta + beta_step * (shape_avg ? mean_F(d_beta) : d_beta)
  int shape_avg;                       // 0 none, 1 mean over all F frames, 2 mean over the valid frames
  const int* seq_lengths = nullptr;    // [T/F], only read when shape_avg == 2
  float* rot;                          // [T][22][9]
  float* feat;                         // [T][200]
  float* out_theta; float* out_beta;   // optional dense copies [T][66], [T][10]
  float* out_theta2; float* out_beta2; // optional second copy (history)
  float* theta_t = nullptr;            // optional: the updated theta in tile layout [tiles][66][64] (smpl_tile.hip)
  int T, F;
  int rod_conv = 0;                    // EMPOSE_RODRIGUES_SMPLX (0) or EMPOSE_RODRIGUES_SO3 (1)
};
hipError_t launch_update_feat(const FeatArgs& a, hipStream_t stream);

struct ChainArgs {
  SmplTables tab;
  const float* rot;        // [T][22][9]
  const float* out;        // [T][ncp] = feat . wc^T
  const float* offset_r;   // [T/F][12][9]
  const float* offset_t;   // [T/F][12][3]
  const float* tgt; int ld_tgt;   // network-input layout; nullptr => forward only
  const float* frame_scale;       // [T]
  int n_markers; int used_slot[12];  // used_slot[m] = column slot of virtual sensor m in tgt, or -1
  float* pos; float* ori; float* joints;   // [T][36], [T][108], [T][66]
  float* pos2; float* ori2; float* joints2;  // optional second copies
  float* d_out;            // [T][ncp]
  float* d_rot;            // [T][22][9]
  int T, F;
  const float* cot_pos = nullptr;    // [T][36]  external cotangents (vector-Jacobian product for training);
  const float* cot_ori = nullptr;    // [T][108] when set they replace the residual of `tgt`
  const float* cot_joints = nullptr; // [T][66]  optional
};
size_t chain_lds_bytes(const SmplTables& tab, int frames_per_block);
hipError_t launch_chain_sensors(const ChainArgs& a, hipStream_t stream);


struct RodBwdArgs {
  const float* theta; int ld_theta;
  const float* d_rot;      // [T][22][9]
  const float* d_feat;     // [T][200]
  float* g_theta; int ld_g;
  float* g_beta; int ld_gb;
  float* trace_g_theta; float* trace_g_beta;  // optional dense copies
  int T;
  int rod_conv = 0;
};
hipError_t launch_rodrigues_bwd(const RodBwdArgs& a, hipStream_t stream);

// ---------------------------------------------------------------------------------------------------------------
// SMPL sub-mesh with one frame per lane (smpl_tile.hip): tables, arguments, launchers.
// "Tile layout" of a per-frame matrix X[T][C]: X_t[tile = t / 64][c][t % 64] -- a column of 64 frames is contiguous.
// ---------------------------------------------------------------------------------------------------------------
constexpr int TL_FR = 64;             // frames per tile = lanes of a wave
constexpr int ROD_BWD_LD = 77;        // row stride of rodrigues_bwd_tile's staging block [TL_FR][77] (66 g_theta | 10 g_beta | pad)
constexpr int TL_NR = 8;              // largest ring (= sensor vertex degree) the kernel takes
constexpr int TL_NLOC = TL_NR + 1;    // local vertices of a sensor patch: centre + ring
constexpr int TL_NBL = 8;              // distinct bones under one patch
struct TileSensor {
  int deg, helper, col, nb;           // ring size, ring index of the helper vertex, first column of the patch, bones
  int bone[TL_NBL];                   // the patch's bones (unused slots: bone 0 with zero weights)
  int bones, pad[3];                  // bit mask of the bones (host: scheduling)
  float w[TL_NLOC][TL_NBL];           // dense skin weights of local vertex i over the patch's bones
};
struct TileTables {                   // one per model, in device memory, read with scalar loads
  int ncp2, j_off2, nloc, nbl;        // columns of the tile (multiple of 32), first rest-joint column, most local
                                      // vertices / bones of a patch
  int n_rounds;
  TileSensor s[12];
  int parent[22];
};
bool build_tile_tables(int nv, int kb, int max_deg, int j_off, const float* wc, const int* parents, const int* skin_idx,
                       const float* skin_w, const int* s_center, const int* s_helper, const int* s_deg,
                       const int* s_faces, TileTables* out, std::vector<float>* wc2);
struct TileArgs {
  const TileTables* tab;
  const float* theta; int ld_theta;   // [T][ld] axis-angles (66 used)
  const float* theta_t = nullptr;     // the same in tile layout [tiles][66][64], or nullptr (strided loads: 64 cache
                                      // lines per load instruction)
  const float* tgt_t = nullptr;       // tgt in tile layout [tiles][12 * n_markers][64], or nullptr
  const float* out_t;                 // tile layout [tiles][ncp2][64]: v_posed of the patches | rest joints
  const float* offset_r;              // [T/F][12][9]
  const float* offset_t;              // [T/F][12][3]
  const float* tgt; int ld_tgt;       // network-input layout; nullptr with backward => external cotangents
  const float* frame_scale;           // [T]
  int n_markers; int used_slot[12];
  float* pos; float* ori; float* joints;       // [T][36], [T][108], [T][66] or nullptr (not wanted)
  float* pos2; float* ori2; float* joints2;    // optional second copies
  float* d_out_t;                     // tile layout [tiles][ncp2][64]
  float* d_rot_t;                     // tile layout [tiles][198][64]
  int T, F;
  int rod_conv;
  const float* cot_pos = nullptr;     // [T][36], [T][108]: external cotangents instead of the residual
  const float* cot_ori = nullptr;
};
hipError_t launch_smpl_tile(const TileArgs& a, bool backward, int nloc, int nbl, hipStream_t stream);
struct RodBwdTArgs {
  const float* theta; int ld_theta;
  const float* theta_t = nullptr;     // optional: the same values in tile layout [tiles][66][64] (coalesced loads)
  const float* d_rot_t;               // tile layout [tiles][198][64]
  const float* d_feat_t; int ld_feat_t;   // tile layout [tiles][ld_feat_t][64] (columns 0..198 used)
  float* g_theta; int ld_g; float* g_beta; int ld_gb;
  float* trace_g_theta; float* trace_g_beta;
  int T; int rod_conv;
};
hipError_t launch_rodrigues_bwd_t(const RodBwdTArgs& a, hipStream_t stream);
hipError_t launch_rows_to_tile(const float* src, int ld, int cols, float* dst_t, int T, hipStream_t stream);
// C = A . Wp^T with tile-layout operands (mlp_fused.hip): A row-major [M][lda] or tile layout (a_tile), C in tile layout
// [tiles][ldc_t][64] (ldc_t == N rounded up to 32; the padding columns are written as zeros).
hipError_t launch_gemm_rows_t(const float* A, int lda, bool a_tile, const float* Wp, float* C_t, int ldc_t, int M, int N,
                              int K, hipStream_t stream);

// The same two products with the small kernels around them folded in (mlp_fused.hip): the pose / shape update +
// feature row as the prologue of the first (the [T][200] feature matrix never reaches HBM), the Rodrigues reverse as
// the epilogue of the second (neither do the feature cotangents).
// x3: Wp = three bf16 pieces per weight in bf16-MFMA fragment order (pack_fragments_x3_raw), product on the bf16 matrix path
hipError_t launch_blend_feat_gemm(const FeatArgs& fa, const float* Wp, float* C_t, int ldc_t, int N, bool x3,
                                  hipStream_t stream);
hipError_t launch_blend_t_gemm_rod(const float* A_t, int lda_t, const float* Wp, int K, const RodBwdTArgs& ra, bool x3,
                                   hipStream_t stream);

// The two init heads on the LSTM output as one product over their stacked columns (mlp_fused.hip): Wp = the stacked weight
// [n_pose + n_shape][K] in fragment order, bias stacked likewise.
bool heads_rows_applicable(int M, int K);
hipError_t launch_heads_rows(const float* y, int ldy, const float* Wp, const float* bias, float* theta, int ld_theta,
                             float* shape, int ld_shape, int M, int K, int n_pose, int n_shape, bool x3, hipStream_t stream);

// Full-mesh: chain only (joints + relative transforms) and dense skinning.
constexpr int MESH_MAX_JOINTS = 52;   // SMPL-H: 22 body + 2 x 15 hand joints
struct MeshChainArgs {
  const float* rot; const float* out; int ncp; int j_off;   // out: rest joints [T][ncp], n_joints * 3 used from j_off
  const int* parents;      // [n_joints], topologically ordered
  const float* trans;      // [T][3] or nullptr
  float* xf;               // [T][22][3][4]: per body bone and row (G^R[r][0..2], A^t[r])
  float* joints;           // [T][n_joints][3]
  int T;
  int n_joints = NB;       // 22 (body) or up to MESH_MAX_JOINTS (hand joints ride on their parents, zero hand pose)
};
hipError_t launch_mesh_chain(const MeshChainArgs& a, hipStream_t stream);
struct MeshSkinArgs {
  const float* feat;           // [T][200]
  const float* wc;             // [V*3 (+66)][200]: row 3s+c = coefficients of coordinate c of vertex s
  const float* xf;             // [T][22][3][4]: row r of bone b = (G^R[r][0..2], A^t[r])
  const int* skin_idx; const float* skin_w; int kb;
  const float* trans;
  float* vertices;             // [T][V][3]
  int T, V;
  const float* wc_frag;        // [tiles][25][3][64][4]: wc in fragment order per 32-vertex tile (api.hip pack_mesh_tiles)
  const int* skin_idx4; const float* skin_w4;   // [tiles * 32][4]
  const void* wc_bf16 = nullptr;   // bf16 pieces of wc in fragment order per tile (api.hip pack_mesh_tiles_bf16), or nullptr
  const void* skin_bf16 = nullptr; // dense skin weights per 32-vertex tile as bf16 pieces in fragment order (pack_mesh_skin_bf16)
  const void* wc_x3 = nullptr;     // THREE bf16 pieces of wc in fragment order per tile (api.hip pack_mesh_tiles_x3)
  int stagger = 1;                 // mesh_rows_x3_kernel: a tile's stores ride in the next tile's K loop, at a per-wave k-step
};
hipError_t launch_mesh_rows(const MeshSkinArgs& a, hipStream_t stream);
// The same evaluation with the blend-shape contraction in split bf16 (mesh.hip); needs MeshSkinArgs::wc_bf16.
hipError_t launch_mesh_rows_bf16(const MeshSkinArgs& a, hipStream_t stream);
constexpr int MESH_BF16_TILE_BYTES = 14 * 3 * 2 * 1024;
// ... and with the bone blend as a second contraction on the matrix cores (needs MeshSkinArgs::skin_bf16 too).
hipError_t launch_mesh_rows_bf16s(const MeshSkinArgs& a, hipStream_t stream);
constexpr int MESH_SKIN_BF16_TILE_BYTES = 2 * 2 * 1024;
// The blend-shape contraction on three bf16 pieces per operand -- fp32-equivalent, the default arithmetic of the full-mesh
// evaluation since round 6 (mesh_x3.hip; option mesh_x3: 1 = products, skinning, stores one after the other with the stores
// staggered into the next tile's K loop, 2 = the same with the stores at the end of their tile, 3 = the skinning
// software-pipelined under the next tile's K loop (measured no faster), 0 = the fp32 MFMA kernel).  Four bones per vertex
// (kb <= 4); needs MeshSkinArgs::wc_x3.
hipError_t launch_mesh_rows_x3(const MeshSkinArgs& a, bool overlap, hipStream_t stream);
constexpr int MESH_X3_TILE_BYTES = 13 * 9 * 1024;

struct VirtualSensorArgs {
  const float* vertices;   // [T][V][3]
  const int* center; const int* helper; const int* deg; const int* faces;  // [M], [M], [M], [M][max_deg][3] (mesh ids)
  float* pos; float* ori; float* normals;   // [T][M][3], [T][M][9], [T][M][3] (un-normalised) or nullptr
  int T, V, M, max_deg;
};
hipError_t launch_virtual_sensors(const VirtualSensorArgs& a, hipStream_t stream);

struct MetricsArgs {
  const float* joints_gt; const float* joints_hat;   // [T][22][3]
  const float* pose_gt; const float* pose_hat;       // [T][63] body axis-angles (no root) or nullptr
  double* rows;                                      // [T][65] = 22 distances | 22 aligned distances | 21 angles (deg)
  int parents[22];
  int T;
};
hipError_t launch_metrics_rows(const MetricsArgs& a, hipStream_t stream);

}  // namespace empose
