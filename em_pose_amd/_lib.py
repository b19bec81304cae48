"""
ctypes binding of the C ABI declared in include/empose_hip.h (libempose_hip.so, built in-tree by em_pose_amd/build.py).

There is no fallback: if the shared library is missing the import of any compute entry point raises, and every
compute call insists on CUDA(ROCm)-resident tensors.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (EMPOSE_LIB_PATH: a lab build of the library, scripts/dev/*_lab.sh -- development only; the default is the in-tree build)
LIB_PATH = os.environ.get('EMPOSE_LIB_PATH') or os.path.join(_HERE, 'csrc', 'libempose_hip.so')

MAX_DENSE = 8
RODRIGUES = {'smplx': 0, 'so3': 1}   # EMPOSE_RODRIGUES_* (include/empose_hip.h)
c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)


class SmplDesc(C.Structure):
    _fields_ = [('n_sensors', C.c_int), ('nv', C.c_int), ('j_off', C.c_int), ('ncp', C.c_int), ('kb', C.c_int),
                ('max_deg', C.c_int),
                ('wc', c_float_p), ('wct', c_float_p), ('parents', c_int_p),
                ('skin_idx', c_int_p), ('skin_w', c_float_p),
                ('bone_ptr', c_int_p), ('bone_vert', c_int_p), ('bone_w', c_float_p),
                ('s_center', c_int_p), ('s_helper', c_int_p), ('s_deg', c_int_p), ('s_faces', c_int_p),
                ('path_ptr', c_int_p), ('path', c_int_p), ('sub_ptr', c_int_p), ('sub', c_int_p),
                ('rodrigues', C.c_int)]


class DenseDesc(C.Structure):
    _fields_ = [('in_dim', C.c_int), ('out_dim', C.c_int), ('weight', c_float_p), ('bias', c_float_p),
                ('bn_weight', c_float_p), ('bn_bias', c_float_p), ('bn_mean', c_float_p), ('bn_var', c_float_p),
                ('bn_eps', C.c_float), ('has_prelu', C.c_int), ('prelu', C.c_float)]


class MlpDesc(C.Structure):
    _fields_ = [('n_layers', C.c_int), ('skip', C.c_int), ('layers', DenseDesc * MAX_DENSE)]


class LstmDesc(C.Structure):
    _fields_ = [('num_layers', C.c_int), ('input_size', C.c_int), ('hidden_size', C.c_int),
                ('w_ih', c_float_p * 4), ('w_hh', c_float_p * 4), ('b_ih', c_float_p * 4), ('b_hh', c_float_p * 4)]


class RnnDesc(C.Structure):
    _fields_ = [('num_layers', C.c_int), ('input_size', C.c_int), ('hidden_size', C.c_int), ('bidirectional', C.c_int),
                ('w_ih', c_float_p * 8), ('w_hh', c_float_p * 8), ('b_ih', c_float_p * 8), ('b_hh', c_float_p * 8)]


class ModelDesc(C.Structure):
    _fields_ = [('smpl', SmplDesc), ('n_markers', C.c_int), ('marker_idx', C.c_int * 12),
                ('n_iterations', C.c_int), ('step_size', C.c_float), ('shape_avg', C.c_int),
                ('use_gradient', C.c_int), ('rnn_init', C.c_int),
                ('rnn', LstmDesc), ('pose_head', DenseDesc), ('shape_head', DenseDesc),
                ('pose_init', MlpDesc), ('shape_init', MlpDesc), ('pose_iter', MlpDesc), ('shape_iter', MlpDesc)]


class LgdIO(C.Structure):
    _fields_ = [('B', C.c_int), ('F', C.c_int),
                ('marker_pos', C.c_void_p), ('marker_oris', C.c_void_p), ('offset_t', C.c_void_p),
                ('offset_r', C.c_void_p), ('marker_masks', C.c_void_p), ('seq_lengths', C.c_void_p),
                ('h0', C.c_void_p), ('c0', C.c_void_p), ('h_n', C.c_void_p), ('c_n', C.c_void_p),
                ('pose_hat', C.c_void_p), ('shape_hat', C.c_void_p), ('joints_hat', C.c_void_p),
                ('hist_pose', C.c_void_p), ('hist_shape', C.c_void_p), ('hist_joints', C.c_void_p),
                ('hist_markers', C.c_void_p), ('hist_markers_ori', C.c_void_p),
                ('trace_g_pose', C.c_void_p), ('trace_g_shape', C.c_void_p),
                ('suppress_missing', C.c_int), ('mask_value', C.c_float)]


class LstmParams(C.Structure):   # empose_lstm_params: DEVICE pointers
    _fields_ = [('num_layers', C.c_int), ('input_size', C.c_int), ('hidden_size', C.c_int),
                ('w_ih', C.c_void_p * 4), ('w_hh', C.c_void_p * 4), ('b_ih', C.c_void_p * 4), ('b_hh', C.c_void_p * 4)]


class LstmGrads(C.Structure):    # empose_lstm_grads
    _fields_ = [('w_ih', C.c_void_p * 4), ('w_hh', C.c_void_p * 4), ('b_ih', C.c_void_p * 4), ('b_hh', C.c_void_p * 4),
                ('d_h0', C.c_void_p * 4), ('d_c0', C.c_void_p * 4)]


class MlpParams(C.Structure):    # empose_mlp_params: DEVICE pointers
    _fields_ = [('n_layers', C.c_int), ('in_dim', C.c_int), ('hidden', C.c_int), ('out_dim', C.c_int),
                ('weight', C.c_void_p * MAX_DENSE), ('bias', C.c_void_p * MAX_DENSE),
                ('bn_weight', C.c_void_p * MAX_DENSE), ('bn_bias', C.c_void_p * MAX_DENSE),
                ('bn_running_mean', C.c_void_p * MAX_DENSE), ('bn_running_var', C.c_void_p * MAX_DENSE),
                ('bn_num_batches', C.c_void_p * MAX_DENSE), ('prelu', C.c_void_p * MAX_DENSE),
                ('bn_eps', C.c_float), ('bn_momentum', C.c_float), ('weight_t', C.c_void_p * MAX_DENSE),
                ('save_layout', C.c_int),
                ('weight_x3', C.c_void_p * MAX_DENSE), ('weight_t_x3', C.c_void_p * MAX_DENSE)]


class MlpGrads(C.Structure):     # empose_mlp_grads
    _fields_ = [('weight', C.c_void_p * MAX_DENSE), ('bias', C.c_void_p * MAX_DENSE),
                ('bn_weight', C.c_void_p * MAX_DENSE), ('bn_bias', C.c_void_p * MAX_DENSE),
                ('prelu', C.c_void_p * MAX_DENSE)]


class LossIO(C.Structure):       # empose_loss_io
    _fields_ = [('B', C.c_int), ('F', C.c_int), ('n_hist', C.c_int), ('n_markers', C.c_int),
                ('marker_idx', C.c_int * 12),
                ('pose_hist', C.c_void_p), ('shape_hist', C.c_void_p), ('markers_hist', C.c_void_p),
                ('markers_ori_hist', C.c_void_p), ('joints_final', C.c_void_p), ('pose_gt', C.c_void_p),
                ('shape_gt', C.c_void_p), ('joints_gt', C.c_void_p), ('inputs', C.c_void_p), ('ld_inputs', C.c_int),
                ('seq_lengths', C.c_void_p), ('marker_masks', C.c_void_p),
                ('w_pose', C.c_float), ('w_shape', C.c_float), ('w_fk', C.c_float), ('w_rec', C.c_float),
                ('d_pose', C.c_void_p), ('d_shape', C.c_void_p), ('d_markers', C.c_void_p),
                ('d_markers_ori', C.c_void_p), ('d_joints', C.c_void_p), ('loss_vals', C.c_void_p)]


class MeshDesc(C.Structure):
    _fields_ = [('n_vertices', C.c_int), ('j_off', C.c_int), ('ncp', C.c_int), ('kb', C.c_int),
                ('wc', c_float_p), ('skin_idx', c_int_p), ('skin_w', c_float_p), ('parents', c_int_p),
                ('n_joints', C.c_int), ('rodrigues', C.c_int), ('with_bf16x3', C.c_int)]


# symbol -> (restype, argtypes); this table is also what tests check against the header.
SIGNATURES = {
    'empose_last_error': (C.c_char_p, []),
    'empose_version': (C.c_int, []),
    'empose_arch': (C.c_char_p, []),
    'empose_set_option': (C.c_int, [C.c_char_p, C.c_int]),
    'empose_get_option': (C.c_int, [C.c_char_p]),
    'empose_reset_options': (C.c_int, []),
    'empose_async_status': (C.c_int, []),
    'empose_pack_weight_x3_bytes': (C.c_size_t, [C.c_int, C.c_int]),
    'empose_pack_weight_x3': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'empose_model_create': (C.c_int, [C.POINTER(ModelDesc), C.POINTER(C.c_void_p)]),
    'empose_model_destroy': (None, [C.c_void_p]),
    'empose_lgd_workspace_bytes': (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    'empose_lgd_forward': (C.c_int, [C.c_void_p, C.POINTER(LgdIO), C.c_void_p, C.c_size_t, C.c_void_p]),
    'empose_lgd_forward_phase': (C.c_int, [C.c_void_p, C.POINTER(LgdIO), C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]),
    'empose_smpl_workspace_bytes': (C.c_size_t, [C.c_void_p, C.c_int]),
    'empose_smpl_sensors_fwd_bwd': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                               C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                               C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    'empose_smpl_vjp_workspace_bytes': (C.c_size_t, [C.c_void_p, C.c_int]),
    'empose_smpl_sensors_vjp': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'empose_update_workspace_bytes': (C.c_size_t, [C.c_void_p, C.c_int]),
    'empose_update_nets_fwd': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_size_t, C.c_void_p]),
    'empose_lstm_workspace_bytes': (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    'empose_lstm_fwd': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                   C.c_void_p]),
    'empose_linear_f32': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p]),
    'empose_linear_f32_ex': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float,
                                        C.c_void_p]),
    'empose_gemm_strided_applicable': (C.c_int, [C.c_int, C.c_int]),
    'empose_gemm_strided_f32': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_long,
                                           C.c_long, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'empose_bn_prelu_workspace_bytes': (C.c_size_t, [C.c_int, C.c_int]),
    'empose_bn_prelu_train_fwd': (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'empose_bn_prelu_train_bwd': (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_size_t, C.c_void_p]),
    'empose_gemm_atb_workspace_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'empose_gemm_atb_f32': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                       C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'empose_transpose_f32': (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    'empose_mlp_train_save_layout': (C.c_int, [C.POINTER(MlpParams), C.c_int]),
    'empose_mlp_train_save_floats': (C.c_size_t, [C.POINTER(MlpParams), C.c_int]),
    'empose_mlp_train_workspace_bytes': (C.c_size_t, [C.POINTER(MlpParams), C.c_int]),
    'empose_mlp_train_fwd': (C.c_int, [C.POINTER(MlpParams), C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'empose_mlp_train_bwd': (C.c_int, [C.POINTER(MlpParams), C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                        C.c_void_p, C.POINTER(MlpGrads), C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    'empose_lgd_assemble_inputs': (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_int, C.c_void_p]),
    'empose_lgd_additive_update': (C.c_int, [C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'empose_lgd_cotangent_step': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float,
                                           C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'empose_mlp_train_stash_floats': (C.c_size_t, [C.POINTER(MlpParams), C.c_int]),
    'empose_mlp_train_bwd_deferred': (C.c_int, [C.POINTER(MlpParams), C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                               C.c_void_p, C.POINTER(MlpGrads), C.c_int, C.c_void_p, C.c_void_p,
                                               C.c_size_t, C.c_void_p]),
    'empose_mlp_train_uses_weight_t': (C.c_int, [C.POINTER(MlpParams), C.c_int]),
    'empose_mlp_train_pair_workspace_bytes': (C.c_size_t, [C.POINTER(MlpParams), C.POINTER(MlpParams), C.c_int]),
    'empose_mlp_train_fwd_pair': (C.c_int, [C.POINTER(MlpParams), C.POINTER(MlpParams), C.c_int, C.c_void_p, C.c_int,
                                           C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_size_t, C.c_void_p]),
    'empose_mlp_train_bwd_deferred_pair': (C.c_int, [C.POINTER(MlpParams), C.POINTER(MlpParams), C.c_int, C.c_void_p,
                                                    C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                                    C.c_void_p, C.POINTER(MlpGrads), C.POINTER(MlpGrads), C.c_int,
                                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'empose_mlp_train_wgrad_workspace_bytes': (C.c_size_t, [C.POINTER(MlpParams), C.c_int, C.c_int]),
    'empose_mlp_train_wgrad': (C.c_int, [C.POINTER(MlpParams), C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_int,
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(MlpGrads), C.c_int,
                                        C.c_void_p, C.c_size_t, C.c_void_p]),
    'empose_smpl_tile_supported': (C.c_int, [C.c_void_p]),
    'empose_pack_inputs': (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'empose_window_mean': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    'empose_axpby2d': (C.c_int, [C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int,
                                  C.c_void_p, C.c_int, C.c_void_p]),
    'empose_lgd_losses_workspace_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'empose_lgd_losses': (C.c_int, [C.POINTER(LossIO), C.c_void_p, C.c_size_t, C.c_void_p]),
    'empose_adam_step': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p]),
    'empose_lstm_train_save_floats': (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'empose_lstm_train_workspace_bytes': (C.c_size_t, [C.POINTER(LstmParams), C.c_int, C.c_int]),
    'empose_lstm_train_fwd': (C.c_int, [C.POINTER(LstmParams), C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_size_t, C.c_void_p]),
    'empose_lstm_train_bwd': (C.c_int, [C.POINTER(LstmParams), C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(LstmGrads), C.c_void_p,
                                         C.c_size_t, C.c_void_p]),
    'empose_rnn_create': (C.c_int, [C.POINTER(RnnDesc), C.POINTER(C.c_void_p)]),
    'empose_rnn_destroy': (None, [C.c_void_p]),
    'empose_rnn_workspace_bytes': (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    'empose_rnn_fwd': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'empose_virtual_sensors_fwd': (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'empose_profile_enable': (C.c_int, [C.c_int]),
    'empose_profile_enable_only': (C.c_int, [C.c_char_p]),
    'empose_profile_ntags': (C.c_int, []),
    'empose_profile_tag_name': (C.c_char_p, [C.c_int]),
    'empose_profile_read': (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    'empose_profile_gemm_kernel_name': (C.c_char_p, [C.c_int] * 5),
    'empose_mesh_create': (C.c_int, [C.POINTER(MeshDesc), C.POINTER(C.c_void_p)]),
    'empose_mesh_destroy': (None, [C.c_void_p]),
    'empose_mesh_workspace_bytes': (C.c_size_t, [C.c_void_p, C.c_int]),
    'empose_mesh_n_joints': (C.c_int, [C.c_void_p]),
    'empose_mesh_joints_fwd': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_size_t, C.c_void_p]),
    'empose_metrics_rows': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int),
                                       C.c_void_p, C.c_void_p]),
    'empose_mesh_vertices_fwd': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'empose_mesh_vertices_fwd_bf16x3': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
}

_lib = None


class EmposeError(RuntimeError):
    pass


def lib():
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EmposeError('HIP extension not built: {} is missing. Run `python em_pose_amd/build.py` '
                              '(or __graft_entry__.build()). There is no CPU fallback.'.format(LIB_PATH))
        # PyTorch first: the library links libamdhip64.so.7 by name, and the process must end up with ONE HIP runtime --
        # the one torch ships and allocates device memory with.  Loaded the other way round, the library binds the
        # system copy under /opt/rocm and torch brings its own: two runtimes, and this one then sees no device
        # ("no ROCm-capable device is detected" from the first hipMalloc; seen with build() + smoke() in one process).
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        raise EmposeError('libempose_hip: error {}: {}'.format(rc, lib().empose_last_error().decode()))


def fptr(arr):
    """Host pointer of a C-contiguous float32 numpy array (kept alive by the caller)."""
    assert arr.dtype == np.float32 and arr.flags['C_CONTIGUOUS']
    return arr.ctypes.data_as(c_float_p)


def iptr(arr):
    assert arr.dtype == np.int32 and arr.flags['C_CONTIGUOUS']
    return arr.ctypes.data_as(c_int_p)


def dptr(t):
    """Device pointer of a torch tensor (None -> NULL). The tensor must live on the GPU and be contiguous."""
    if t is None:
        return None
    if not t.is_cuda:
        raise EmposeError('the HIP path needs tensors on the GPU (got device {}); there is no CPU fallback'
                          .format(t.device))
    if not t.is_contiguous():
        raise EmposeError('tensor must be contiguous')
    return C.c_void_p(t.data_ptr())


def current_stream():
    """Handle of torch's current stream.  Every launch of this package goes onto it, which is also what makes
    temporaries safe without host synchronisation: the caching allocator hands a freed block only to later work on
    the same stream, i.e. to work that is ordered after the kernels still reading it."""
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class option(object):
    """Context manager: `with _lib.option('gemm_splitk', 0): ...` selects a kernel variant (empose_set_option) and
    restores the previous value."""

    def __init__(self, name, value):
        self.name, self.value = name.encode(), int(value)

    def __enter__(self):
        self.prev = lib().empose_get_option(self.name)
        check(lib().empose_set_option(self.name, self.value))
        return self

    def __exit__(self, *exc):
        check(lib().empose_set_option(self.name, self.prev))
        return False


def profile_read():
    """{tag name: (total_ms, launches)} since the last read; see empose_profile_read."""
    l = lib()
    n = l.empose_profile_ntags()
    ms = (C.c_double * n)()
    cnt = (C.c_longlong * n)()
    check(l.empose_profile_read(ms, cnt))
    return {l.empose_profile_tag_name(i).decode(): (ms[i], cnt[i]) for i in range(n) if cnt[i] > 0}
