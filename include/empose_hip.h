/*
 * empose_hip.h -- C ABI of the MI355X (gfx950) implementation of EM-POSE's learned-gradient-descent fitting loop.
 *
 * The reference (facebookresearch/em-pose) has no FFI/operator layer for this path: it sits behind Python objects
 * (SURVEY.md 8b).  This header is the boundary the build introduces underneath those objects; every entry point
 * names the reference interface it replaces.  Conventions:
 *   - extern "C", plain C types only; every function returns 0 on success or a negative EMPOSE_E* code, and
 *     empose_last_error() returns a thread-local description.  No exception crosses the boundary.
 *   - All tensors are caller-owned DEVICE pointers, fp32, row-major contiguous unless a leading dimension is given;
 *     index data is int32.  Model descriptors (empose_*_desc) hold HOST pointers and are consumed at create time.
 *   - Every compute call takes a stream handle (a hipStream_t passed as void*; NULL = default stream), is
 *     asynchronous and stream-ordered, and never allocates: scratch comes from the caller via
 *     empose_lgd_workspace_bytes().  A model is immutable after creation, so concurrent calls on distinct streams
 *     with distinct workspaces are safe.
 */
#ifndef EMPOSE_HIP_H_
#define EMPOSE_HIP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMPOSE_OK 0
#define EMPOSE_EINVAL (-1)   /* bad argument / unsupported configuration */
#define EMPOSE_EHIP (-2)     /* a HIP runtime call failed */
#define EMPOSE_ENOMEM (-3)   /* workspace too small / allocation failed */
#define EMPOSE_ETIMEOUT (-4) /* a cooperative kernel gave up waiting for another workgroup: outputs poisoned with NaN */

#define EMPOSE_N_BODY 22      /* root + 21 body joints, reference configuration.py:104 */
#define EMPOSE_N_SENSORS 12   /* virtual sensors always evaluated, reference models.py:383,538 */
#define EMPOSE_K_FEAT 200     /* 189 pose-feature + 10 shape + 1 */
#define EMPOSE_MAX_DENSE 8
#define EMPOSE_N_JOINTS_SMPLH 52  /* joints of `body.Jtr`, reference bodymodels/smpl.py:121-122 */

/* How the axis-angle -> rotation map guards the angle at r = 0.  The body-model arithmetic lives in an un-vendored
 * dependency (human_body_prior fork, reference requirements.txt:9), so the convention is a property of the model
 * handle instead of being baked into the kernels:
 *   SMPLX  angle = ||r + 1e-8||                (smplx lbs.batch_rodrigues; default)
 *   SO3    angle = sqrt(clamp(||r||^2, 1e-4))  (reference helpers/so3.py:116-121, so3_exponential_map) */
#define EMPOSE_RODRIGUES_SMPLX 0
#define EMPOSE_RODRIGUES_SO3 1

typedef void* empose_stream_t;
typedef struct empose_model empose_model_t;
typedef struct empose_mesh empose_mesh_t;
typedef struct empose_rnn empose_rnn_t;

const char* empose_last_error(void);
/* Library/ABI version and the offload architecture it was compiled for ("gfx950"). */
int empose_version(void);
const char* empose_arch(void);

/* Kernel-variant selection for A/B measurements and bit-identity tests: several paths have two implementations (the
 * one-launch fused update MLPs vs layer-by-layer GEMMs, whole-sequence LSTM kernels vs step launches, ...) that must
 * give the same results.  Options are process-wide ints, default 1 = the faster variant; the library never reads the
 * environment.  Names: "mlp_fused", "lstm_persist", "gemm_splitk", "gemm_wide", "atb_target", "atb_chunk",
 * "smpl_tile" (frame-per-lane SMPL sub-mesh kernels: 0 never, 1 from 16384 frames on [default], 2 always),
 * "smpl_fuse" (on that path: pose / shape update and Rodrigues reverse inside the blend GEMMs; 0 = own kernels),
 * "heads_rows" (both init heads as one row-block product from 4096 frames on; 0 = two problems on the generic tile),
 * "lstm_seq" (batches above 256 rows: the LSTM's whole sequence in one cooperative launch; 0 [default] = one launch per
 * wavefront step, which measured faster), "bptt_wave" (training: reverse LSTM recurrences of two layers as a wavefront),
 * "train_fused" (train-mode MLP layer with BatchNorm / PReLU folded into the GEMMs: 0 never [default], 1 above 1024
 * rows, 2 always), "train_epi" (train-mode MLP layer with the BatchNorm statistics in the GEMM epilogues and ONE
 * combine-and-apply launch per layer and direction: 0 never, 1 above 1024 rows [default], 2 always), "atb_fast" (weight
 * gradients: whole-tile / whole-chunk products on a branch-free interior kernel, bit-identical; 0 = the general kernel),
 * "mlp_x3" (the fused update MLPs form every fp32 product from three bf16 pieces per operand -- six bf16 matrix-core
 * products with fp32 accumulation, fp32-equivalent and 2.7 times the fp32 instruction's rate -- when every hidden width
 * is a multiple of 64: 1 [default]; 0 = the fp32 MFMA instruction; 2 = a variant whose waves share the operand split
 * through LDS, measured slower), "lstm_x3" (the same arithmetic for the LSTM steps of batches above 256 rows, inference,
 * uni-directional stacks with a hidden size of whole 32s: 1 [default]; 0 = the fp32 MFMA instruction), "rows_x3" (the
 * same arithmetic for the row-block products with fused prologue / epilogue: the blend products of the frame-per-lane
 * SMPL path and the stacked init heads; 0 = the fp32 MFMA instruction), "train_cols" (training at up to 512 rows: a
 * layer's product + BatchNorm + PReLU as one launch, forward and backward, both update networks side by side -- see
 * empose_mlp_train_fwd_pair; 0 = a product and a BatchNorm launch per layer and network), "lstm_fewrows" (LSTM steps of 4 to
 * 16 rows -- the reference's training batch, chunks of recordings -- as launches whose workgroups own two hidden units and
 * split K over all their threads; 0 = the whole-sequence kernel / the small-batch step kernel, which share their bits),
 * "mesh_skin_mfma" (split-bf16 full-mesh variant only: the bone blend as a second matrix-core contraction; 0 [default,
 * measured faster] = vector skinning), "spin_limit" (see empose_async_status).
 * empose_get_option returns -1 for an unknown name.  New in this library (no counterpart in the reference). */
int empose_set_option(const char* name, int value);
int empose_get_option(const char* name);
int empose_reset_options(void);   /* every option back to its default */

/* Launches are asynchronous, so a failure INSIDE a kernel cannot be the return value of the call that launched it.  The
 * one such failure this library has: the whole-sequence LSTM kernels (small batches: lstm_persist; opt-in lstm_seq) are
 * cooperative -- workgroups poll exchange words written by other workgroups -- and a poll that exceeds its spin limit
 * gives up, writes NaN from there on (it never hangs the GPU) and counts itself in a host-visible word.
 *   - every later empose_lstm_fwd / empose_rnn_fwd / empose_lgd_forward[_phase] call of the process returns
 *     EMPOSE_ETIMEOUT instead of running (count in empose_last_error()) -- STICKY, whichever model, stream or thread it
 *     belongs to -- until
 *   - empose_async_status() has returned EMPOSE_ETIMEOUT once (it reports and clears; EMPOSE_OK otherwise): call it after
 *     synchronising the stream, before trusting / averaging outputs (em_pose_amd.eval.helpers, bench.py and
 *     scripts/train.py do), and re-create the state the failed call carried (it is NaN).
 * Option "spin_limit" (> 0) forces the limit of those polls (tests).  The reference's forward has no counterpart (a
 * single Python thread, reference nn/layers.py:133-157). */
int empose_async_status(void);

/* ---- model description (host pointers) ------------------------------------------------------------------------ */

/* Packed SMPL-H constants for the sensor sub-mesh; produced by em_pose_amd/bodymodels/tables.py.
 * Replaces what the reference keeps as BodyModel buffers + VirtualMarkerHelper index caches
 * (reference bodymodels/smpl.py:42-67, data/virtual_sensors.py:47-75). */
typedef struct {
  int n_sensors;          /* 12 */
  int nv;                 /* needed vertices */
  int j_off;              /* column of the first rest-joint coordinate in `wc` rows */
  int ncp;                /* padded row count of wc (multiple of 4) */
  int kb;                 /* skinning weights per vertex (after folding the hands into the wrists) */
  int max_deg;            /* faces per sensor vertex (padded) */
  const float* wc;        /* [ncp][200] */
  const float* wct;       /* [200][ncp] */
  const int* parents;     /* [22] */
  const int* skin_idx;    /* [nv][kb] */
  const float* skin_w;    /* [nv][kb] */
  const int* bone_ptr;    /* [23] CSR over bones */
  const int* bone_vert;   /* [bone_ptr[22]] */
  const float* bone_w;    /* [bone_ptr[22]] */
  const int* s_center;    /* [12] local vertex of the sensor */
  const int* s_helper;    /* [12] local helper vertex */
  const int* s_deg;       /* [12] */
  const int* s_faces;     /* [12][max_deg][3] local vertex ids */
  const int* path_ptr;    /* [23] root->joint paths */
  const int* path;
  const int* sub_ptr;     /* [23] subtree member lists */
  const int* sub;
  int rodrigues;          /* EMPOSE_RODRIGUES_*: honoured by the forward, the residual gradient and the training VJP */
} empose_smpl_desc;

/* One Linear (+ BatchNorm1d in eval mode) (+ PReLU): reference nn/layers.py:13-43,46-77. weight is [out][in]. */
typedef struct {
  int in_dim, out_dim;
  const float* weight;
  const float* bias;
  const float* bn_weight; /* NULL => no batch norm */
  const float* bn_bias;
  const float* bn_mean;
  const float* bn_var;
  float bn_eps;
  int has_prelu;
  float prelu;            /* single shared slope, nn.PReLU() */
} empose_dense_desc;

/* reference nn/layers.py:46-77: layers[0]=input_to_hidden, then 2*num_blocks hidden layers, then hidden_to_output. */
typedef struct {
  int n_layers;           /* 2 + 2*num_blocks, <= EMPOSE_MAX_DENSE; 0 => MLP absent */
  int skip;               /* m_skip_connections: residual around every 2-layer block */
  empose_dense_desc layers[EMPOSE_MAX_DENSE];
} empose_mlp_desc;

/* reference nn/layers.py:114 (nn.LSTM, unidirectional, gate order i,f,g,o). */
typedef struct {
  int num_layers;         /* <= 4; 0 => absent */
  int input_size, hidden_size;
  const float* w_ih[4];   /* [4H][in] */
  const float* w_hh[4];   /* [4H][H] */
  const float* b_ih[4];
  const float* b_hh[4];
} empose_lstm_desc;

/* reference nn/models.py:372-394,424-457 (IterativeErrorFeedback.__init__/create_model). */
typedef struct {
  empose_smpl_desc smpl;
  int n_markers;          /* 6 or 12: sensors fed to the networks */
  int marker_idx[12];     /* first n_markers entries: which of the 12 virtual sensors they are (S_CONFIG_6) */
  int n_iterations;       /* N */
  float step_size;
  int shape_avg;          /* 0 off; 1 mean over all F frames incl. padding (reference models.py:529-532);
                             2 ragged rows behave as unpadded windows of their own length: shape mean over the
                               valid frames and residual weight 1 instead of F/len (batched streaming evaluation) */
  int use_gradient;
  int rnn_init;
  empose_lstm_desc rnn;           /* rnn_init */
  empose_dense_desc pose_head;    /* rnn_init: Linear(H,66) */
  empose_dense_desc shape_head;   /* rnn_init: Linear(H,10) */
  empose_mlp_desc pose_init;      /* !rnn_init */
  empose_mlp_desc shape_init;
  empose_mlp_desc pose_iter;
  empose_mlp_desc shape_iter;
} empose_model_desc;

/* Replaces IterativeErrorFeedback construction + load_state_dict (reference models.py:23-33, eval/helpers.py:131-137):
 * copies and packs every constant to the current HIP device. */
int empose_model_create(const empose_model_desc* desc, empose_model_t** out);
void empose_model_destroy(empose_model_t* model);

/* ---- the whole N-step loop ------------------------------------------------------------------------------------ */

/* Inputs are what `batch.get_inputs()` yields (reference data/data.py:304-309,433-459). */
typedef struct {
  int B, F;                      /* windows x frames; T = B*F */
  const float* marker_pos;       /* [B][F][12*3] */
  const float* marker_oris;      /* [B][F][12*9] row-major 3x3 */
  const float* offset_t;         /* [B][12][3] */
  const float* offset_r;         /* [B][12][3][3] */
  const float* marker_masks;     /* [B][F][12] 1=present, or NULL */
  const int* seq_lengths;        /* [B] or NULL (= F everywhere) */
  const float* h0;               /* [L][B][H] carried LSTM state or NULL (zeros): reference layers.py:149-150 */
  const float* c0;
  float* h_n;                    /* [L][B][H] out, may be NULL */
  float* c_n;
  /* outputs (reference models.py:602-609,631): pose_hat holds root+body (66); callers slice [..., :3] / [..., 3:] */
  float* pose_hat;               /* [B][F][66] */
  float* shape_hat;              /* [B][F][10] */
  float* joints_hat;             /* [B][F][66] */
  /* optional histories, N+1 entries each (reference models.py:620-629); NULL to skip */
  float* hist_pose;              /* [N+1][T][66] */
  float* hist_shape;             /* [N+1][T][10] */
  float* hist_joints;            /* [N+1][T][66] */
  float* hist_markers;           /* [N+1][T][12*3] */
  float* hist_markers_ori;       /* [N+1][T][12*9] */
  /* optional trace of the gradient features fed to the update nets (reference models.py:578-582); NULL to skip */
  float* trace_g_pose;           /* [N][T][66] */
  float* trace_g_shape;          /* [N][T][10] */
  /* RealBatch.get_inputs replaces the readings of missing sensors by a constant before the model sees them
   * (reference data/data.py:284-302,304-309).  suppress_missing != 0: marker_pos / marker_oris are the RAW readings
   * and the packing kernel does that replacement (x * valid + mask_value * !valid with valid = mask == 1). */
  int suppress_missing;
  float mask_value;
} empose_lgd_io;

size_t empose_lgd_workspace_bytes(const empose_model_t* model, int B, int F);

/* Replaces IterativeErrorFeedback.forward for one window batch (reference models.py:485-632). */
int empose_lgd_forward(const empose_model_t* model, const empose_lgd_io* io, void* workspace, size_t workspace_bytes,
                       empose_stream_t stream);

/* The same forward in two parts that may be issued on different streams: EMPOSE_LGD_PHASE_INIT = input packing + the
 * initial estimate (LSTM with state carry + heads, or the init MLPs; reads io->h0/c0, writes io->h_n/c_n),
 * EMPOSE_LGD_PHASE_ITER = the N refinement iterations and the outputs.  The second part reads what the first left in
 * `workspace` (same io, same workspace; order them with an event).  A streaming caller runs chunk c + 1's INIT -- which
 * depends only on chunk c's final LSTM state -- beside chunk c's ITER, with two workspaces.  New in this library; the
 * reference runs the chunks of a recording strictly one after the other (scripts/evaluate_real.py:39-61). */
#define EMPOSE_LGD_PHASE_INIT 1
#define EMPOSE_LGD_PHASE_ITER 2
int empose_lgd_forward_phase(const empose_model_t* model, const empose_lgd_io* io, void* workspace,
                             size_t workspace_bytes, empose_stream_t stream, int phases);

/* ---- building blocks (also what the unit tests drive) --------------------------------------------------------- */

/* One SMPL-H evaluation restricted to the sensor sub-mesh and, optionally, the residual gradient.
 * Replaces get_estimated_real_markers + reconstruction_loss + autograd.backward
 * (reference models.py:471-483,560-579; loss.py:23-41; virtual_sensors.py:85-96).
 *   theta [T][ld_theta] (66 used), beta [T][ld_beta] (10 used)
 *   offset_r/offset_t per WINDOW ([T/F][12][3][3], [T/F][12][3]); F = frames per window
 *   targets: tgt [T][ld_tgt] holding n_markers*3 positions then n_markers*9 orientations (the network input
 *            layout of prepare_inputs, reference models.py:106-125); frame_scale [T] per-frame loss weight
 *   outputs pos [T][36], ori [T][108], joints [T][66]; g_theta [T][ld_g] (66), g_beta [T][ld_gb] (10) or NULL.
 * workspace: empose_smpl_workspace_bytes(model, T).
 * Launches of 16384 frames and more run the frame-per-lane kernels (csrc/smpl_tile.hip) when the model's sensor patches
 * allow it -- closed triangle fans of at most 8 faces over at most 8 bones, what a closed manifold body mesh gives;
 * empose_smpl_tile_supported says whether they do (option "smpl_tile": 0 never, 1 by size, 2 always). */
size_t empose_smpl_workspace_bytes(const empose_model_t* model, int T);
int empose_smpl_tile_supported(const empose_model_t* model);
int empose_smpl_sensors_fwd_bwd(const empose_model_t* model, int T, int F,
                                const float* theta, int ld_theta, const float* beta, int ld_beta,
                                const float* offset_r, const float* offset_t,
                                const float* tgt, int ld_tgt, const float* frame_scale,
                                float* pos, float* ori, float* joints,
                                float* g_theta, int ld_g, float* g_beta, int ld_gb,
                                void* workspace, size_t workspace_bytes, empose_stream_t stream);

/* Vector-Jacobian product of the same evaluation for training: given cotangents of the outputs
 * d_pos [T][36], d_ori [T][108], d_joints [T][66] (or NULL) returns g_theta [T][66], g_beta [T][10].
 * This is the backward of `get_estimated_real_markers` that the reference gets from autograd when
 * IterativeErrorFeedback.backward calls total_loss.backward() (reference models.py:634-688). */
size_t empose_smpl_vjp_workspace_bytes(const empose_model_t* model, int T);
int empose_smpl_sensors_vjp(const empose_model_t* model, int T, int F, const float* theta, int ld_theta,
                            const float* beta, int ld_beta, const float* offset_r, const float* offset_t,
                            const float* d_pos, const float* d_ori, const float* d_joints, float* g_theta,
                            float* g_beta, void* workspace, size_t workspace_bytes, empose_stream_t stream);

/* Both update networks on x [T][ldx]: d_pose [T][66], d_shape [T][10]
 * (replaces pose_net_iter / shape_net_iter, reference models.py:586-587; layers.py:46-77). */
size_t empose_update_workspace_bytes(const empose_model_t* model, int T);
int empose_update_nets_fwd(const empose_model_t* model, int T, const float* x, int ldx, float* d_pose, float* d_shape,
                           void* workspace, size_t workspace_bytes, empose_stream_t stream);

/* The init LSTM over ragged windows: x [B][F][ldx] (input_size used) -> y [B][F][H]
 * (replaces RNNLayer.forward, reference layers.py:133-157). */
size_t empose_lstm_workspace_bytes(const empose_model_t* model, int B, int F);
int empose_lstm_fwd(const empose_model_t* model, int B, int F, const float* x, int ldx, const int* seq_lengths,
                    const float* h0, const float* c0, float* y, float* h_n, float* c_n,
                    void* workspace, size_t workspace_bytes, empose_stream_t stream);

/* C[M][ldc] = act((A[M][lda] . W[N][ldw]^T) * scale[n] + shift[n]) on the fp32 matrix cores; scale/shift may be NULL.
 * Exposed for tests and for host code that needs a plain fp32 linear layer. */
int empose_linear_f32(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K,
                      const float* scale, const float* shift, int prelu, float slope, empose_stream_t stream);

/* As empose_linear_f32 with an optional residual operand resid [M][ldr]:
 *   act 0: y = acc*scale + shift (+ resid);  act 1: PReLU(slope) then + resid (MLP skip connection, layers.py:35-43);
 *   act 2: + resid, then ReLU (FeedForwardResidualBlock of the ResNet baseline, reference nn/layers.py:170-182). */
int empose_linear_f32_ex(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K,
                         const float* scale, const float* shift, const float* resid, int ldr, int act, float slope,
                         empose_stream_t stream);

/* Small GEMM with strided operands: C[M][ldc] = A . W^T (+ bias[n]) where A(m, k) = A[m * a_rs + k * a_ks] and
 * W(n, k) = W[n * w_rs + k * w_ks] -- any of the four transposition cases without a transposed copy.  This is what the
 * backward pass of the training path's linear layers needs (torch.nn.functional.linear's autograd, reference
 * nn/layers.py:46-77 in train mode): dX = dY . W and dW = dY^T . X.  Only problems of at most 512 output tiles of
 * 32 x 32 (empose_gemm_strided_applicable); larger ones belong to a library GEMM. */
int empose_gemm_strided_applicable(int M, int N);
int empose_gemm_strided_f32(int M, int N, int K, const float* A, long a_rs, long a_ks, const float* W, long w_rs,
                            long w_ks, float* C, int ldc, const float* bias, empose_stream_t stream);

/* Train-mode BatchNorm1d + PReLU (one shared slope) of an MLP hidden layer, forward and backward as one kernel each
 * (reference nn/layers.py:13-77 in training mode; torch.nn.BatchNorm1d semantics: batch statistics, biased variance for
 * the normalisation, running statistics updated with `momentum` and the unbiased variance, num_batches_tracked + 1).
 * x, z, dz, dx are [M][ld] row-major; gamma, beta, save_mean, save_rstd, dgamma, dbeta are [C]; slope is one device float;
 * dslope receives the slope gradient; dslope_partial (ceil(C / 32) floats) and counter (one int, zero before the first
 * call, left at zero by every call) are scratch. running_mean / running_var / num_batches_tracked may be NULL.
 * Batches of more than 1024 rows split the rows over workgroups and need empose_bn_prelu_workspace_bytes(M, C) bytes of
 * workspace for the partial sums (0 for smaller batches, workspace may then be NULL). */
size_t empose_bn_prelu_workspace_bytes(int M, int C);
int empose_bn_prelu_train_fwd(int M, int C, const float* x, int ldx, const float* gamma, const float* beta,
                              const float* slope, float eps, float momentum, float* running_mean, float* running_var,
                              long long* num_batches_tracked, float* z, int ldz, float* save_mean, float* save_rstd,
                              void* workspace, size_t workspace_bytes, empose_stream_t stream);
int empose_bn_prelu_train_bwd(int M, int C, const float* x, int ldx, const float* dz, int lddz, const float* gamma,
                              const float* beta, const float* slope, const float* save_mean, const float* save_rstd,
                              float* dx, int lddx, float* dgamma, float* dbeta, float* dslope, float* dslope_partial,
                              int* counter, void* workspace, size_t workspace_bytes, empose_stream_t stream);

/* ---- training backward: building blocks (BASELINE.json configs[4]) ------------------------------------------------ */
/* The reference trains through torch.autograd (reference nn/models.py:634-688, scripts/train.py:133-152); these are the
 * hand-written pieces the Python autograd Functions of em_pose_amd/nn call instead of library kernels.  All pointers are
 * DEVICE pointers (the parameters live in torch tensors and change every optimiser step: nothing is packed or cached). */

/* C[N][ldc] = A[M][lda]^T . B[M][ldb], optionally bias[n] = sum_m A[m][n]: the weight and bias gradient of a linear
 * layer y = x W^T + b (dW = dY^T X, db = column sums of dY; torch.nn.functional.linear's backward).  The reduction over
 * M is split over workgroups and summed in a fixed order (deterministic).  workspace: empose_gemm_atb_workspace_bytes. */
size_t empose_gemm_atb_workspace_bytes(int M, int N, int K);
int empose_gemm_atb_f32(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                        float* bias, void* workspace, size_t workspace_bytes, empose_stream_t stream);
/* dst[c][r] = src[r][c] (rows x cols): W^T copies so that dX = dY . W runs on the forward GEMM (empose_linear_f32). */
int empose_transpose_f32(int rows, int cols, const float* src, int ld_src, float* dst, int ld_dst,
                         empose_stream_t stream);

/* One MLP of the LGD model (reference nn/layers.py:46-77) in TRAINING mode: Linear -> BatchNorm1d (batch statistics) ->
 * PReLU for every layer but the last, which is a plain Linear.  Device pointers to the parameters as torch holds them. */
typedef struct {
  int n_layers;                 /* 2 + 2 * blocks, <= EMPOSE_MAX_DENSE; no skip connections */
  int in_dim, hidden, out_dim;  /* in_dim and hidden multiples of 4 */
  const float* weight[EMPOSE_MAX_DENSE];   /* [out_l][in_l] */
  const float* bias[EMPOSE_MAX_DENSE];
  const float* bn_weight[EMPOSE_MAX_DENSE];   /* layers 0 .. n_layers-2 */
  const float* bn_bias[EMPOSE_MAX_DENSE];
  float* bn_running_mean[EMPOSE_MAX_DENSE];   /* updated in place like torch.nn.BatchNorm1d, may be NULL */
  float* bn_running_var[EMPOSE_MAX_DENSE];
  long long* bn_num_batches[EMPOSE_MAX_DENSE];
  const float* prelu[EMPOSE_MAX_DENSE];       /* one device float per layer */
  float bn_eps, bn_momentum;
  /* Optional, backward only: the weights of layers 1 .. n_layers-1 transposed, [in_l][out_l] with the leading dimension
   * of the last layer padded to a multiple of 4 and its padding columns ZERO (empose_transpose_f32 into a zeroed buffer).
   * With NULL entries the backward transposes the weights itself on every call; a caller that applies the network
   * several times per step (the LGD loop) transposes once per step instead.  empose_mlp_train_uses_weight_t() says
   * whether the backward of M rows reads them at all (the one-launch layers of up to 512 rows read W itself). */
  const float* weight_t[EMPOSE_MAX_DENSE];
  /* How `save` is laid out: 0 = whatever the options "train_fused" / "train_epi" select AT THE TIME OF EACH CALL (every
   * call of a step must then see the same options); 1 / 2 / 3 = the layout empose_mlp_train_save_layout() reported when
   * the forward ran -- the backward and weight-gradient calls of that step then read the buffer the way it was written
   * even if an option changed in between (an A/B script, another thread). */
  int save_layout;
  /* Optional (round 6), large batches only: the weights of layers 0 .. n_layers-2 (weight_x3: W_l itself, [hidden][in_l])
   * and of layers 1 .. n_layers-1 (weight_t_x3: the transposed copy `weight_t[l]`, [in_l][ld]) as three bf16 pieces per
   * weight in matrix-core fragment order, made by empose_pack_weight_x3 once per optimiser step.  When present (and option
   * "train_x3" != 0, M >= 1024, hidden % 64 == 0) the forward products y_l = a_{l-1} W_l^T and the reverse products
   * dA_{l-1} = dY_l W_l form every fp32 product from three bf16 pieces per operand (fp32-equivalent; bf16x3.h) instead of
   * running on the fp32 MFMA instruction.  NULL entries: the fp32 instruction. */
  const unsigned short* weight_x3[EMPOSE_MAX_DENSE];
  const unsigned short* weight_t_x3[EMPOSE_MAX_DENSE];
} empose_mlp_params;
/* A weight matrix W [N][ldw] (K columns used, K % 4 == 0) as three bf16 pieces per weight in fragment order; `out`:
 * empose_pack_weight_x3_bytes(N, K) bytes of device memory, 16-byte aligned. */
size_t empose_pack_weight_x3_bytes(int N, int K);
int empose_pack_weight_x3(const float* W, int ldw, int N, int K, void* out, empose_stream_t stream);
typedef struct {                /* gradient outputs, shapes of the parameters */
  float* weight[EMPOSE_MAX_DENSE];
  float* bias[EMPOSE_MAX_DENSE];
  float* bn_weight[EMPOSE_MAX_DENSE];
  float* bn_bias[EMPOSE_MAX_DENSE];
  float* prelu[EMPOSE_MAX_DENSE];
} empose_mlp_grads;
/* `save`: per hidden layer the pre-BatchNorm and the activated outputs [M][hidden] and the batch mean / rstd. */
/* 1 (BatchNorm kernels) / 2 (BatchNorm folded into the GEMMs) / 3 (statistics in the GEMM epilogues): the layout the
 * current options select for M rows; store it in empose_mlp_params::save_layout for the calls of the step. */
int empose_mlp_train_save_layout(const empose_mlp_params* p, int M);
/* 1: the backward of M rows reads `weight_t` (or transposes on every call when it is NULL); 0: it does not. */
int empose_mlp_train_uses_weight_t(const empose_mlp_params* p, int M);
size_t empose_mlp_train_save_floats(const empose_mlp_params* p, int M);
size_t empose_mlp_train_workspace_bytes(const empose_mlp_params* p, int M);
/* x [M][ldx] -> out [M][ld_out] (out_dim columns written). */
int empose_mlp_train_fwd(const empose_mlp_params* p, int M, const float* x, int ldx, float* out, int ld_out,
                         float* save, void* workspace, size_t workspace_bytes, empose_stream_t stream);
/* d_out [M][ld_dout]: ld_dout a multiple of 4 and >= out_dim, columns past out_dim ZERO.  Parameter gradients are
 * overwritten (accumulate = 0) or added to (1: the same network applied N times in the LGD loop).  The input x is not
 * differentiated: the LGD loop feeds the update networks detached values (reference models.py:549-551). */
int empose_mlp_train_bwd(const empose_mlp_params* p, int M, const float* x, int ldx, const float* d_out, int ld_dout,
                         const float* save, const empose_mlp_grads* grads, int accumulate, void* workspace,
                         size_t workspace_bytes, empose_stream_t stream);

/* Deferred weight gradients (the same network applied n_app times, as the N iterations of the LGD loop): the backward
 * of each application keeps its layer cotangents in `dz_stash` (empose_mlp_train_stash_floats floats) instead of forming
 * dW / db, and ONE call of empose_mlp_train_wgrad forms them over the rows of all applications -- one A^T B product of
 * n_app * M rows per layer instead of n_app products and reductions.  BatchNorm / PReLU parameter gradients are produced
 * by the deferred backward exactly as by empose_mlp_train_bwd.  x / save / dz_stash of wgrad: host arrays of n_app
 * device pointers (n_app <= 8), the arguments the applications were run with.  Same sums in a different order:
 * results agree with the per-application path to rounding, not bitwise.
 * The stash keeps d_out at dz_stash + M * (n_layers - 1) * hidden, row stride (out_dim + 3) & ~3: a caller that
 * produces d_out there (d_out = that address, ld_dout = that stride) saves the copy. */
size_t empose_mlp_train_stash_floats(const empose_mlp_params* p, int M);
int empose_mlp_train_bwd_deferred(const empose_mlp_params* p, int M, const float* x, int ldx, const float* d_out,
                                  int ld_dout, const float* save, const empose_mlp_grads* grads, int accumulate,
                                  float* dz_stash, void* workspace, size_t workspace_bytes, empose_stream_t stream);
/* Both update networks of one LGD iteration (they read the same rows x; reference nn/models.py:584-587 applies
 * `pose_net` and `shape_net` to the same detached input) in one call.  Up to 512 rows -- the reference's training batch of
 * 12 windows x 32 frames (scripts/train.py:125-152) -- every layer of BOTH networks is ONE launch, forward (product +
 * bias + train-mode BatchNorm + PReLU) and backward (dA = dZ W + the BatchNorm / PReLU reverse of the layer below):
 * a workgroup per 16 columns x quarter of the rows, whose column statistics meet through a mailbox of tagged words inside
 * the workspace (csrc/train_cols.hip; option "train_cols", 0 = a product and a BatchNorm launch per layer and network).
 * Networks the paired launches do not cover (different depth / hidden width / BatchNorm constants, more rows) run one
 * after the other through the single-network entry points: same results either way.
 * Reads and writes save layout 1, like empose_mlp_train_fwd / _bwd at these sizes (which use the same launches for one
 * network).  A poll of the mailbox that gives up is reported like the cooperative LSTM kernels' (empose_async_status).
 * Like those kernels the paired launches need every workgroup of a launch resident at once -- the device to themselves:
 * processes that SHARE one GPU must set "train_cols" to 0 (em_pose_amd/helpers/distributed.py does when ranks wrap around
 * the devices), or their launches starve each other until the polls give up; the same holds beside collective kernels that
 * wait for other ranks on another stream (the helper's gradient buckets switch it off for the overlapped sweep).
 * workspace: empose_mlp_train_pair_workspace_bytes. */
size_t empose_mlp_train_pair_workspace_bytes(const empose_mlp_params* p0, const empose_mlp_params* p1, int M);
int empose_mlp_train_fwd_pair(const empose_mlp_params* p0, const empose_mlp_params* p1, int M, const float* x, int ldx,
                              float* out0, int ld_out0, float* out1, int ld_out1, float* save0, float* save1,
                              void* workspace, size_t workspace_bytes, empose_stream_t stream);
int empose_mlp_train_bwd_deferred_pair(const empose_mlp_params* p0, const empose_mlp_params* p1, int M, const float* x,
                                       int ldx, const float* d_out0, int ld_dout0, const float* d_out1, int ld_dout1,
                                       const float* save0, const float* save1, const empose_mlp_grads* grads0,
                                       const empose_mlp_grads* grads1, int accumulate, float* dz_stash0, float* dz_stash1,
                                       void* workspace, size_t workspace_bytes, empose_stream_t stream);
size_t empose_mlp_train_wgrad_workspace_bytes(const empose_mlp_params* p, int n_app, int M);
int empose_mlp_train_wgrad(const empose_mlp_params* p, int n_app, int M, const float* const* x, int ldx,
                           const float* const* save, const float* const* dz_stash, const empose_mlp_grads* grads,
                           int accumulate, void* workspace, size_t workspace_bytes, empose_stream_t stream);

/* Network input rows and the per-frame residual weight as one launch.
 * Replaces `BaseModel.prepare_inputs` (reference models.py:106-125: reshape, 6-sensor subset S_CONFIG_6, concat
 * positions | row-major orientations) and the frame weighting of `reconstruction_loss` as the loop applies it
 * (reference loss.py:31-39 with the B * F rescale of models.py:578-579): weight[t] = [f < len_b] * F / len_b *
 * [all 12 sensors present].
 *   marker_pos [B*F][36], marker_oris [B*F][108]; marker_idx: the n_markers (<= 12) sensors taken, in order;
 *   marker_masks [B*F][12] or NULL; seq_lengths [B] (int32) or NULL (= F everywhere);
 *   x [B*F][ldx]: columns [0, 12 * n_markers) are written; frame_weight [B*F] or NULL (not wanted). */
int empose_pack_inputs(int B, int F, int n_markers, const int* marker_idx, const float* marker_pos,
                       const float* marker_oris, const float* marker_masks, const int* seq_lengths, float* x, int ldx,
                       float* frame_weight, empose_stream_t stream);

/* out[t][c] = mean of in[.][c] over the window of frame t (F consecutive rows); the operator is its own adjoint, so
 * the same call back-propagates (`_to_single_shape`, reference models.py:529-535,588-589). */
int empose_window_mean(int T, int F, int C, const float* in, int ld_in, float* out, int ld_out, empose_stream_t stream);
/* out = alpha x + beta y over a [rows][cols] block with row strides; x or y may be NULL (treated as 0), out may alias. */
int empose_axpby2d(int rows, int cols, float alpha, const float* x, int ldx, float beta, const float* y, int ldy,
                   float* out, int ldo, empose_stream_t stream);

/* Bookkeeping of the LGD loop around the update networks as single launches (each replaces 3-9 axpby / mean launches; at
 * the reference's training batch a step is launch-bound):
 *   assemble_inputs   X[t] = [x0[t] (d_in) | pose[t] (66) | shape[t] (10)]; the gradient columns d_in+76.. are written by
 *                     empose_smpl_sensors_fwd_bwd (reference models.py:584)
 *   additive_update   pose_next = pose + step d_pose, shape_next = shape + step (mean over the window of) d_shape
 *                     (reference models.py:588-600)
 *   cotangent_step    reverse sweep at history entry i: Dp = [Dp +] d_pose + vp [+ g_theta / (B F)], Ds likewise (loss
 *                     terms, body-model VJP, the in-forward E_i.backward() deposit of models.py:576), and with dpad / dspad
 *                     the zero-padded cotangents [T][68] / [T][12] of the update networks' outputs:
 *                     step * Dp, step * (window mean of) Ds.  `first`: Dp / Ds are overwritten, not accumulated. */
int empose_lgd_assemble_inputs(int T, int d_in, const float* x0, int ld_x0, const float* pose, const float* shape,
                               float* X, int ldx, empose_stream_t stream);
int empose_lgd_additive_update(int B, int F, float step, int shape_avg, const float* pose, const float* d_pose,
                               const float* shape, const float* d_shape, float* pose_next, float* shape_next,
                               empose_stream_t stream);
int empose_lgd_cotangent_step(int B, int F, int first, const float* d_pose, const float* d_shape, const float* vp,
                              const float* vs, const float* g_theta, int ld_g, const float* g_beta, int ld_gb, float* Dp,
                              float* Ds, float step, int shape_avg, float* dpad, float* dspad, empose_stream_t stream);

/* The loss of IterativeErrorFeedback.backward (reference models.py:634-688, loss.py:13-41) and the cotangents of the
 * total loss with respect to every history entry, in one pass. */
typedef struct {
  int B, F, n_hist, n_markers;  /* n_hist = N + 1 */
  int marker_idx[12];           /* first n_markers entries: which virtual sensors feed the network */
  const float* pose_hist;       /* [n_hist][T][66] */
  const float* shape_hist;      /* [n_hist][T][10] */
  const float* markers_hist;    /* [n_hist][T][36] */
  const float* markers_ori_hist;/* [n_hist][T][108] */
  const float* joints_final;    /* [T][66] */
  const float* pose_gt;         /* [T][66] */
  const float* shape_gt;        /* [B][10] */
  const float* joints_gt;       /* [T][66] or NULL (no FK loss) */
  const float* inputs; int ld_inputs;   /* network input rows: n_markers*3 positions then n_markers*9 orientations */
  const int* seq_lengths;       /* [B] or NULL */
  const float* marker_masks;    /* [T][12] or NULL */
  float w_pose, w_shape, w_fk, w_rec;
  float* d_pose; float* d_shape; float* d_markers; float* d_markers_ori;   /* like the histories */
  float* d_joints;              /* [T][66] */
  float* loss_vals;             /* [5] device floats: pose, shape, reconstruction, fk, total_loss */
} empose_loss_io;
size_t empose_lgd_losses_workspace_bytes(int B, int F, int n_hist);
int empose_lgd_losses(const empose_loss_io* io, void* workspace, size_t workspace_bytes, empose_stream_t stream);

/* torch.optim.Adam (amsgrad off, no weight decay) over n tensors in one launch.  The pointer tables are DEVICE arrays:
 * params/grads/exp_avg/exp_avg_sq [n] (void*), sizes [n] (int64), and the chunk tables chunk_tensor [n_chunks] (int32),
 * chunk_offset [n_chunks] (int64): chunk c covers elements [offset, offset + 4096) of tensor chunk_tensor[c].
 * step is the 1-based step count (bias correction). */
int empose_adam_step(int n_chunks, const void* params, const void* grads, const void* exp_avg, const void* exp_avg_sq,
                     const void* sizes, const void* chunk_tensor, const void* chunk_offset, float lr, float beta1,
                     float beta2, float eps, int step, empose_stream_t stream);

/* Stacked uni-directional LSTM with autograd support (reference nn/layers.py:133-157 in training mode: nn.LSTM over
 * packed ragged sequences, its backward through cuDNN/MIOpen).  Parameters as torch.nn.LSTM holds them. */
typedef struct {
  int num_layers, input_size, hidden_size;   /* num_layers <= 4 */
  const float* w_ih[4];   /* [4H][in]  (in = input_size for layer 0, H above) */
  const float* w_hh[4];   /* [4H][H] */
  const float* b_ih[4];   /* [4H] */
  const float* b_hh[4];
} empose_lstm_params;
typedef struct {          /* gradient outputs, same shapes; every weight / bias pointer required */
  float* w_ih[4];
  float* w_hh[4];
  float* b_ih[4];
  float* b_hh[4];
  /* optional (NULL: not wanted): the cotangents of the initial state, [B][H] per layer -- what a learned initial state
   * (reference nn/layers.py:121-131, `learn_init_state`) is trained with */
  float* d_h0[4];
  float* d_c0[4];
} empose_lstm_grads;
/* floats of the `save` buffer the forward fills for the backward: per layer gates [B][F][4H], cell states [B][F][H],
 * incoming hidden states [B][F][H], output sequence [B][F][H] */
size_t empose_lstm_train_save_floats(int num_layers, int B, int F, int hidden_size);
size_t empose_lstm_train_workspace_bytes(const empose_lstm_params* p, int B, int F);
/* x [B][F][ldx], seq_lengths [B] or NULL, h0/c0 [L][B][H] or NULL -> y [B][F][H], h_n/c_n [L][B][H] (may be NULL). */
int empose_lstm_train_fwd(const empose_lstm_params* p, int B, int F, const float* x, int ldx, const int* seq_lengths,
                          const float* h0, const float* c0, float* y, float* h_n, float* c_n, float* save,
                          void* workspace, size_t workspace_bytes, empose_stream_t stream);
/* dy [B][F][H] -> parameter gradients (overwritten), if dx != NULL dx [B][F][input_size], and where grads->d_h0 / d_c0 are
 * given the cotangents of the initial state.  The final state is not differentiated (it only seeds the next chunk,
 * detached, reference models.py:489-492). */
int empose_lstm_train_bwd(const empose_lstm_params* p, int B, int F, const float* x, int ldx, const int* seq_lengths,
                          const float* c0, const float* save, const float* dy, float* dx,
                          const empose_lstm_grads* grads, void* workspace, size_t workspace_bytes,
                          empose_stream_t stream);

/* ---- stand-alone (Bi)LSTM: the RNNLayer of the BiRNN baseline --------------------------------- */
/* reference nn/layers.py:80-157 (nn.LSTM, optionally bidirectional, packed ragged sequences). Parameter index
 * u = layer * dirs + direction (direction 1 = reverse), as PyTorch orders `*_l{k}` / `*_l{k}_reverse`; layer k > 0 of
 * a bidirectional stack takes 2*hidden inputs. State tensors are [num_layers*dirs][B][H]; y is [B][F][dirs*H]. */
typedef struct {
  int num_layers, input_size, hidden_size, bidirectional;
  const float* w_ih[8];
  const float* w_hh[8];
  const float* b_ih[8];
  const float* b_hh[8];
} empose_rnn_desc;
int empose_rnn_create(const empose_rnn_desc* desc, empose_rnn_t** out);
void empose_rnn_destroy(empose_rnn_t* rnn);
size_t empose_rnn_workspace_bytes(const empose_rnn_t* rnn, int B, int F);
int empose_rnn_fwd(const empose_rnn_t* rnn, int B, int F, const float* x, int ldx, const int* seq_lengths,
                   const float* h0, const float* c0, float* y, float* h_n, float* c_n, void* workspace,
                   size_t workspace_bytes, empose_stream_t stream);

/* Virtual sensor positions and local frames from full-mesh vertices [T][V][3]
 * (replaces VirtualMarkerHelper.get_virtual_pos_and_rot, reference data/virtual_sensors.py:85-96, and
 * compute_vertex_and_face_normals restricted to the sensor vertices, helpers/utils.py:126-146).
 * Index tables are DEVICE int32 arrays in mesh numbering: center/helper/deg [M], faces [M][max_deg][3].
 * Outputs pos [T][M][3], ori [T][M][3][3] (columns tangent, bitangent, normal), normals [T][M][3] un-normalised or NULL. */
int empose_virtual_sensors_fwd(int T, int V, const float* vertices, int M, int max_deg, const int* center,
                               const int* helper, const int* deg, const int* faces, float* pos, float* ori,
                               float* normals, empose_stream_t stream);

/* ---- optional per-launch timing ------------------------------------------------------------------------------- */
/* While enabled, every kernel launch issued by the entry points above is bracketed by HIP events on the launch stream
 * and attributed to one of empose_profile_ntags() categories (GEMMs by role, LSTM step, chain kernel, ...).
 * empose_profile_read() waits for the recorded events, returns per-category total milliseconds and launch counts, and
 * clears the log. New in this library (the reference only has wall-clock prints, scripts/train.py:136-173). */
int empose_profile_enable(int on);
/* Single-kernel mode: only the launches of category `tag_name` are bracketed (start event, end event right after the
 * launch), the rest of the step runs unperturbed; every empose_lgd_forward additionally records one empty event pair,
 * reported as category "event_pair": what two event packets cost by themselves, to be subtracted from the bracketed
 * duration. This is how bench.py times the dominant kernel inside the step. */
int empose_profile_enable_only(const char* tag_name);
int empose_profile_ntags(void);
const char* empose_profile_tag_name(int tag);
int empose_profile_read(double* total_ms, long long* count);
/* Name, as rocprofv3 prints it, of the kernel a linear layer of `count` (1 or 2) problems of shape M x N x K is
 * dispatched to (role 1 = update-net hidden layer); lets bench.py label its roofline entry with the kernel that ran. */
const char* empose_profile_gemm_kernel_name(int M, int N, int K, int count, int role);

/* ---- full-mesh evaluation (final vertices; ground-truth preprocessing) ---------------------------------------- */

typedef struct {
  int n_vertices, j_off, ncp, kb;
  const float* wc;        /* [ncp][200] rows: V*3 vertex coordinates then n_joints*3 rest-joint coordinates */
  const int* skin_idx;    /* [V][kb], bone ids < 22 (hand weights folded into the wrists) */
  const float* skin_w;    /* [V][kb] */
  const int* parents;     /* [n_joints], topologically ordered, parents[0] < 0 */
  int n_joints;           /* posed joints returned: 22 (body) ... 52 (all of SMPL-H, as `body.Jtr`); 0 means 22 */
  int rodrigues;          /* EMPOSE_RODRIGUES_* */
  int with_bf16x3;        /* also pack the TWO-piece split-bf16 tables of empose_mesh_vertices_fwd_bf16x3 (+18 MB); the
                           * three-piece tables of the default path (25 MB) are always packed when kb <= 4 */
} empose_mesh_desc;

int empose_mesh_create(const empose_mesh_desc* desc, empose_mesh_t** out);
void empose_mesh_destroy(empose_mesh_t* mesh);
size_t empose_mesh_workspace_bytes(const empose_mesh_t* mesh, int T);
int empose_mesh_n_joints(const empose_mesh_t* mesh);
/* Replaces SMPLLayer.forward/fk (reference bodymodels/smpl.py:81-147): poses [T][66] (root first), betas [T][10],
 * trans [T][3] or NULL -> vertices [T][V][3], joints [T][n_joints][3] (n_joints = 52 reproduces `body.Jtr`; the 30 hand
 * joints have zero pose, reference smpl.py:99, so they ride rigidly on the wrists' frames).
 * Arithmetic (round 6): fp32 operands and fp32 accumulation; for body models with at most four bones per vertex the
 * blend-shape contraction forms every fp32 product from three bf16 pieces per operand on the bf16 matrix cores
 * (csrc/mesh_x3.hip: fp32-equivalent, measured against float64 in tests/test_hip_round6.py); option "mesh_x3" = 0 selects
 * the kernel on the fp32 MFMA instruction, which models with more bones per vertex always take. */
int empose_mesh_vertices_fwd(const empose_mesh_t* mesh, int T, const float* poses, const float* betas,
                             const float* trans, float* vertices, float* joints,
                             void* workspace, size_t workspace_bytes, empose_stream_t stream);

/* The same evaluation with the blend-shape contraction in split bf16 on the bf16 matrix cores, fp32 accumulate (pose
 * blend-shapes: 2 pieces / 3 products; template + shape: 3 pieces / 6 products; kinematic chain and skinning unchanged,
 * fp32).  NOT the arithmetic of the reference or of the headline benchmark: an explicitly selected variant whose vertices
 * stay within 1e-4 of the fp32 path (tests/test_hip_boundary.py); joints are identical (the chain does not use it). */
int empose_mesh_vertices_fwd_bf16x3(const empose_mesh_t* mesh, int T, const float* poses, const float* betas,
                                    const float* trans, float* vertices, float* joints, void* workspace,
                                    size_t workspace_bytes, empose_stream_t stream);

/* Joints only (forward kinematics without the mesh): what MetricsEngine needs from `smpl_model.fk`
 * (reference eval/metrics.py:223-228 keeps `kp3d[:, :22]` and discards the vertices). joints [T][n_joints][3]; same
 * workspace as above. */
int empose_mesh_joints_fwd(const empose_mesh_t* mesh, int T, const float* poses, const float* betas,
                           const float* trans, float* joints, void* workspace, size_t workspace_bytes,
                           empose_stream_t stream);

/* ---- evaluation metrics ------------------------------------------------------------------------ */
/* Per frame: 22 Euclidean joint distances, 22 distances after similarity-Procrustes alignment of the prediction onto
 * the ground truth, and 21 geodesic angles (degrees) between global joint orientations with the root fixed to the
 * identity. Replaces MetricsEngine._compute_eucl_dist / _procrustes / _compute_angular_dist + local_to_global
 * (reference eval/metrics.py:18-66,110-162; helpers/utils.py:165-199). joints_* [T][22][3], pose_* [T][63] (body
 * axis-angles, no root) or NULL (angles reported as 0), parents: HOST int[22]; rows: device double [T][65]. */
int empose_metrics_rows(int T, const float* joints_gt, const float* joints_hat, const float* pose_gt,
                        const float* pose_hat, const int* parents_host, double* rows, empose_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* EMPOSE_HIP_H_ */
