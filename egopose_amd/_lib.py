"""ctypes binding of libegopose_hip.so (the C-ABI in include/egopose_hip.h).

The product path has no CPU fallback: if the shared library is missing or a symbol cannot be
resolved this module raises, loudly, at import of the first function that needs it.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libegopose_hip.so")

EGP_OK = 0
EGP_EXPERT_ROW = 168

c_int_p = C.POINTER(C.c_int32)
c_dbl_p = C.POINTER(C.c_double)
vp = C.c_void_p


class ModelDesc(C.Structure):
    _fields_ = [
        ("nq", C.c_int32), ("nv", C.c_int32), ("nu", C.c_int32), ("nbody", C.c_int32), ("nM", C.c_int32),
        ("body_qpos_start", c_int_p), ("body_ndof", c_int_p), ("dof_parentid", c_int_p), ("dof_Madr", c_int_p),
        ("ee_body", c_int_p),
        ("jkp", c_dbl_p), ("jkd", c_dbl_p), ("a_ref", c_dbl_p), ("a_scale", c_dbl_p), ("torque_lim", c_dbl_p),
        ("b_diffw", c_dbl_p),
        ("sub_dt", C.c_double), ("frame_skip", C.c_int32), ("episode_len", C.c_int32),
        ("w_p", C.c_double), ("w_v", C.c_double), ("w_e", C.c_double), ("w_rp", C.c_double), ("w_rv", C.c_double),
        ("k_p", C.c_double), ("k_v", C.c_double), ("k_e", C.c_double), ("k_rh", C.c_double), ("k_rq", C.c_double),
        ("k_rl", C.c_double), ("k_ra", C.c_double),
        ("v_ord", C.c_double), ("decay", C.c_int32),
        ("obs_heading", C.c_int32), ("obs_keep_root_heading", C.c_int32), ("obs_coord_root", C.c_int32), ("obs_vel", C.c_int32),
        ("action_torque", C.c_int32), ("obs_phase", C.c_int32),
    ]


class ExpertTable(C.Structure):
    _fields_ = [
        ("n_takes", C.c_int32), ("n_frames", C.c_int32), ("take_offset", c_int_p),
        ("qpos", c_dbl_p), ("qvel", c_dbl_p), ("rlinv_local", c_dbl_p), ("rangv", c_dbl_p), ("rq_rmh", c_dbl_p),
        ("ee_pos", c_dbl_p), ("bquat", c_dbl_p), ("bangvel", c_dbl_p), ("head_height_lb", c_dbl_p),
    ]


class SurrogateDesc(C.Structure):
    _fields_ = [
        ("nq", C.c_int32), ("nv", C.c_int32), ("nu", C.c_int32), ("nbody", C.c_int32), ("nM", C.c_int32),
        ("njoint", C.c_int32),
        ("qM0", c_dbl_p), ("Minv0", c_dbl_p), ("body_parent", c_int_p), ("body_pos", c_dbl_p), ("body_ndof", c_int_p),
        ("joint_axis", c_dbl_p), ("joint_anchor", c_dbl_p),
        ("sub_dt", C.c_double), ("damping", C.c_double), ("support_k", C.c_double), ("support_c", C.c_double),
    ]


class DynamicsDesc(C.Structure):
    _fields_ = [("nbody", C.c_int32), ("njoint", C.c_int32), ("body_parent", c_int_p), ("body_pos", c_dbl_p), ("body_com", c_dbl_p),
                ("body_inertia", c_dbl_p), ("body_mass", c_dbl_p), ("body_ndof", c_int_p), ("joint_axis", c_dbl_p),
                ("joint_anchor", c_dbl_p), ("armature", C.c_double), ("gravity", C.c_double * 3)]


class MlpLayer(C.Structure):
    _fields_ = [("wt", vp), ("bias", vp), ("in_dim", C.c_int32), ("out_dim", C.c_int32)]


class GemmDesc(C.Structure):
    _fields_ = [("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("A", vp), ("lda", C.c_int64), ("a_kcontig", C.c_int32),
                ("B", vp), ("ldb", C.c_int64), ("b_kcontig", C.c_int32),
                ("C", vp), ("ldc", C.c_int64),
                ("bias", vp), ("relu", C.c_int32),
                ("mask", vp), ("ldmask", C.c_int64),
                ("terms", C.c_int32), ("splits", C.c_int32), ("accumulate", C.c_int32),
                ("bias_grad", vp), ("workspace", vp),
                ("a_rows", vp), ("A2", vp), ("lda2", C.c_int64), ("a_split", C.c_int32), ("a_src_rows", C.c_int64),
                ("b_krows", vp), ("B2", vp), ("ldb2", C.c_int64), ("b_split", C.c_int32), ("b_src_rows", C.c_int64),
                ("c_rows", vp), ("a_krows", vp)]


class PpoLossDesc(C.Structure):
    """egp_ppo_loss_desc (include/egopose_hip.h)."""
    _fields_ = [("n", C.c_int32), ("n_pol", C.c_int32), ("act_dim", C.c_int32),
                ("rows", vp), ("pred", vp), ("returns", vp),
                ("mean", vp), ("ld_mean", C.c_int64),
                ("actions", vp), ("ld_act", C.c_int64),
                ("log_std", vp), ("adv", vp),
                ("fixed_logp", vp), ("write_fixed", C.c_int32),
                ("clip_eps", C.c_double), ("inv_n_val", C.c_double), ("inv_n_exp", C.c_double),
                ("d_pred", vp), ("d_mean", vp), ("ld_dmean", C.c_int64), ("d_log_std", vp),
                ("losses", vp), ("workspace", vp)]


class AdamSegment(C.Structure):
    """egp_adam_segment (include/egopose_hip.h)."""
    _fields_ = [("begin", C.c_int64), ("end", C.c_int64),
                ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double), ("weight_decay", C.c_double),
                ("bias1", C.c_double), ("bias2", C.c_double), ("max_norm", C.c_double),
                ("clip_group", C.c_int32)]


ADAM_MAX_SEGMENTS = 8


class RolloutTick(C.Structure):
    """egp_rollout_tick (include/egopose_hip.h): field order and types must match the header."""
    _fields_ = [("ctx", vp), ("eng", vp), ("stream", vp),
                ("n_env", C.c_int32), ("nmax", C.c_int32), ("obs_dim", C.c_int32), ("nu", C.c_int32), ("nq", C.c_int32), ("nv", C.c_int32),
                ("ctx_dim", C.c_int32), ("ctx_T", C.c_int32), ("episode_len", C.c_int32), ("reward_job", C.c_int32),
                ("has_fix_head_lb", C.c_int32),
                ("end_reward", C.c_double), ("zf_clip", C.c_double), ("fix_head_lb", C.c_double),
                ("cur_t", vp), ("frame_base", vp), ("e_ind", vp), ("s_ind", vp), ("steps_done", vp),
                ("active", vp), ("active_i32", vp), ("head_z", vp), ("head_lb", vp),
                ("rec_valid", vp), ("rec_done", vp), ("rec_e_ind", vp), ("rec_s_ind", vp),
                ("states", vp), ("next_states", vp), ("actions", vp), ("rewards", vp), ("cinfo", vp),
                ("noise", vp), ("v_out", vp), ("v_stride", C.c_int64),
                ("layers", vp), ("n_layers", C.c_int32), ("activation", C.c_int32), ("log_std", vp),
                ("slab_host", vp), ("slab_dev", vp),
                ("qpos", vp), ("qvel", vp), ("prev_qpos", vp), ("ee", vp),
                ("zf_workspace", vp), ("reset_scratch", vp),
                ("defer_apply", C.c_int32)]


class HostProbeResult(C.Structure):
    """egp_host_probe_result (include/egopose_hip.h)."""
    _fields_ = [(k, C.c_double) for k in ("pcie_read_gbps", "pcie_read_us_per_pass", "go_rtt_us_p50", "go_rtt_us_p99", "go_rtt_us_max",
                                          "spin_gap_us_max", "spin_gap_us_median_of_thread_max", "spin_lost_frac")] + \
               [(k, C.c_int32) for k in ("pcie_read_rows", "go_rtt_n", "go_in_vram", "large_bar", "spin_threads", "spin_gaps_over_5us")]


class EngineDesc(C.Structure):
    _fields_ = [("n_env", C.c_int32), ("n_threads", C.c_int32), ("n_groups", C.c_int32), ("device_dynamics", C.c_int32)]


PHYS_RESET = C.CFUNCTYPE(C.c_int, vp, C.c_int32, c_dbl_p, c_dbl_p)
PHYS_STEP = C.CFUNCTYPE(C.c_int, vp, C.c_int32, c_dbl_p)
PHYS_DRAIN = C.CFUNCTYPE(C.c_int, vp, C.c_int32, c_dbl_p, c_dbl_p, c_dbl_p, c_dbl_p, c_dbl_p)
PHYS_DESTROY = C.CFUNCTYPE(None, vp)
PHYS_EPOCH = C.CFUNCTYPE(C.c_int64, vp, C.c_int32)


class PhysicsVtable(C.Structure):
    _fields_ = [("user", vp), ("reset", PHYS_RESET), ("step", PHYS_STEP), ("drain", PHYS_DRAIN),
                ("destroy", PHYS_DESTROY), ("name", C.c_char_p), ("inertia_epoch", PHYS_EPOCH)]


# name -> (restype, argtypes); must list every symbol include/egopose_hip.h declares
_i32, _i64, _f64 = C.c_int32, C.c_int64, C.c_double
SIGNATURES = {
    "egp_last_error": (C.c_char_p, []),
    "egp_version": (C.c_char_p, []),
    "egp_abi_sizeof": (C.c_int64, [C.c_char_p]),
    "egp_obs_dim": (_i32, [vp]),
    "egp_create": (C.c_int, [C.POINTER(ModelDesc), C.c_int, C.POINTER(vp)]),
    "egp_destroy": (C.c_int, [vp]),
    "egp_set_reward_weights": (C.c_int, [vp, C.POINTER(ModelDesc)]),
    "egp_set_pd_variant": (C.c_int, [vp, C.c_int]),
    "egp_upload_experts": (C.c_int, [vp, C.POINTER(ExpertTable)]),
    "egp_reward_simple_f64": (C.c_int, [vp, _i32, vp, vp, vp, vp, C.c_double, _i32, vp, vp, vp]),
    "egp_quat_op_f64": (C.c_int, [_i32, vp, vp, _i32, vp, vp]),
    "egp_quat_op_f32": (C.c_int, [_i32, vp, vp, _i32, vp, vp]),
    "egp_body_quat_f64": (C.c_int, [vp, vp, _i32, vp, vp]),
    "egp_body_quat_f32": (C.c_int, [vp, vp, _i32, vp, vp]),
    "egp_obs_f64": (C.c_int, [vp, vp, vp, vp, _i32, vp, vp]),
    "egp_obs_f32": (C.c_int, [vp, vp, vp, vp, _i32, vp, vp]),
    "egp_pd_torque_f64": (C.c_int, [vp, vp, vp, vp, vp, vp, _i32, vp, vp, vp]),
    "egp_pd_torque_f32": (C.c_int, [vp, vp, vp, vp, vp, vp, _i32, vp, vp, vp]),
    "egp_reward_quat_v3_f64": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, _f64, _i32, vp, vp, vp]),
    "egp_reward_quat_v3_f32": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, _f64, _i32, vp, vp, vp]),
    "egp_pose_features_f64": (C.c_int, [vp, vp, vp, vp, _i32, _i32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "egp_pose_features_f32": (C.c_int, [vp, vp, vp, vp, _i32, _i32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "egp_zfilter_workspace_bytes": (_i64, [_i32, _i32]),
    "egp_zfilter_f64": (C.c_int, [vp, vp, _i32, _i32, vp, vp, _i32, _f64, vp, vp, vp]),
    "egp_zfilter_f32": (C.c_int, [vp, vp, _i32, _i32, vp, vp, _i32, _f64, vp, vp, vp]),
    "egp_obs_zfilter_f64": (C.c_int, [vp, vp, vp, vp, vp, _i32, vp, vp, _f64, vp, vp, _i32, vp, vp]),
    "egp_obs_zfilter_f32": (C.c_int, [vp, vp, vp, vp, vp, _i32, vp, vp, _f64, vp, vp, _i32, vp, vp]),
    "egp_gae_workspace_bytes": (_i64, [_i32]),
    "egp_gae_f64": (C.c_int, [vp, vp, vp, _i32, _f64, _f64, vp, vp, vp, vp, vp]),
    "egp_gae_f32": (C.c_int, [vp, vp, vp, _i32, _f64, _f64, vp, vp, vp, vp, vp]),
    "egp_gae_standardize_f64": (C.c_int, [vp, _i32, vp, vp]),
    "egp_gae_standardize_f32": (C.c_int, [vp, _i32, vp, vp]),
    "egp_lstm_gate_layout": (_i32, []),
    "egp_gather_concat_f32": (C.c_int, [vp, _i64, vp, vp, _i64, _i32, _i32, _i32, vp, _i64, vp]),
    "egp_scatter_rows_f32": (C.c_int, [vp, _i64, vp, _i32, _i32, vp, _i64, vp]),
    "egp_gemm_workspace_floats": (_i64, [_i32, _i32, _i32, _i32]),
    "egp_gemm_f32": (C.c_int, [C.POINTER(GemmDesc), vp]),
    "egp_ppo_loss_workspace_bytes": (_i64, [_i32, _i32, _i32]),
    "egp_ppo_loss_f32": (C.c_int, [C.POINTER(PpoLossDesc), vp]),
    "egp_adam_workspace_bytes": (_i64, []),
    "egp_adam_step_f32": (C.c_int, [_i32, C.POINTER(AdamSegment), vp, vp, vp, vp, vp, vp, vp]),
    "egp_adam_step_f64": (C.c_int, [_i32, C.POINTER(AdamSegment), vp, vp, vp, vp, vp, vp, vp, vp]),
    "egp_adam_step_f64g": (C.c_int, [_i32, C.POINTER(AdamSegment), vp, vp, vp, vp, vp, vp, vp]),
    "egp_lstm_fwd_f32": (C.c_int, [vp, vp, _i32, _i32, _i32, _i32, vp, vp, vp, vp]),
    "egp_lstm_bwd_f32": (C.c_int, [vp, vp, vp, vp, _i32, _i32, _i32, _i32, vp, vp]),
    "egp_lstm_group_fwd_f32": (C.c_int, [vp, vp, _i32, _i32, _i32, _i32, _i32, C.POINTER(C.c_void_p), _i32, vp, vp, vp]),
    "egp_lstm_group_bwd_f32": (C.c_int, [C.POINTER(C.c_void_p), _i32, vp, vp, vp, _i32, _i32, _i32, _i32, _i32, vp, vp, vp]),
    "egp_rollout_tick_pre": (C.c_int, [vp, _i32, _i32, _i32, _i32, vp, _i32, vp, vp]),
    "egp_rollout_tick_apply": (C.c_int, [vp, _i32, _i32, _i32, _i32, vp, vp]),
    "egp_obs_zfilter_split_max_rows": (C.c_int32, []),
    "egp_obs_zfilter_stats_f64": (C.c_int, [vp, vp, vp, vp, vp, _i32, vp, vp]),
    "egp_obs_zfilter_apply_f64": (C.c_int, [vp, vp, vp, vp, _i32, vp, vp, C.c_double, vp, vp, vp, vp]),
    "egp_policy_gaussian_filter_f32": (C.c_int, [vp, vp, C.c_int64, _i32, vp, vp, vp, vp, _i32, vp, vp, C.c_double, vp, vp, vp,
                                                vp, _i32, _i32, vp, vp, vp, vp, vp, vp, C.c_int64, vp]),
    "egp_rollout_tick_post": (C.c_int, [vp, _i32, _i32, _i32, _i32, vp, vp, vp, vp]),
    "egp_rollout_reset": (C.c_int, [vp, _i32, _i32, _i32, _i32, vp, _i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "egp_engine_group_stream": (vp, [vp, _i32]),
    "egp_debug_burn": (C.c_int, [C.c_int64, _i32, vp, vp]),
    "egp_lstm_group_fwd_len_f32": (C.c_int, [vp, vp, _i32, _i32, _i32, _i32, _i32, C.POINTER(C.c_void_p), _i32, vp, vp, vp, vp, _i32, vp, vp]),
    "egp_lstm_group_bwd_len_f32": (C.c_int, [C.POINTER(C.c_void_p), _i32, vp, vp, vp, _i32, _i32, _i32, _i32, _i32, vp, vp, vp, vp, _i32, vp]),
    "egp_set_dynamics_model": (C.c_int, [vp, C.POINTER(DynamicsDesc)]),
    "egp_dynamics_f64": (C.c_int, [vp, vp, vp, _i32, vp, C.c_int64, vp, vp, vp]),
    "egp_mlp_pack_floats": (C.c_int64, [_i32, _i32]),
    "egp_mlp_pack_f32": (C.c_int, [vp, C.c_int64, _i32, _i32, vp, vp]),
    "egp_policy_gaussian_f32": (C.c_int, [vp, C.c_int64, _i32, vp, vp, _i32, _i32, C.POINTER(MlpLayer), _i32, _i32, vp, vp, vp, vp, vp]),
    "egp_policy_gaussian_staged_f32": (C.c_int, [vp, C.c_int64, _i32, vp, vp, _i32, _i32, C.POINTER(MlpLayer), _i32, _i32, vp, vp, vp, vp, vp, vp,
                                                 C.c_int64, vp]),
    "egp_physics_register": (C.c_int, [C.POINTER(PhysicsVtable), _i32, C.POINTER(vp)]),
    "egp_physics_create_surrogate": (C.c_int, [C.POINTER(SurrogateDesc), _i32, C.POINTER(vp)]),
    "egp_physics_destroy": (C.c_int, [vp]),
    "egp_physics_name": (C.c_char_p, [vp]),
    "egp_physics_n_env": (_i32, [vp]),
    "egp_physics_reset_host": (C.c_int, [vp, _i32, vp, vp]),
    "egp_physics_step_host": (C.c_int, [vp, _i32, vp]),
    "egp_physics_drain_host": (C.c_int, [vp, _i32, vp, vp, vp, vp, vp]),
    "egp_engine_create": (C.c_int, [vp, vp, C.POINTER(EngineDesc), C.POINTER(vp)]),
    "egp_engine_destroy": (C.c_int, [vp]),
    "egp_engine_state": (C.c_int, [vp] + [C.POINTER(vp)] * 7),
    "egp_engine_reset": (C.c_int, [vp, vp, _i32, vp, vp, vp]),
    "egp_engine_step_async": (C.c_int, [vp, _i32, vp, vp, vp]),
    "egp_engine_wait": (C.c_int, [vp, _i32, vp]),
    "egp_engine_timing": (C.c_int, [vp, c_dbl_p, c_dbl_p, c_dbl_p, C.POINTER(_i64)]),
    "egp_engine_reset_timing": (C.c_int, [vp]),
    "egp_engine_inertia_uploads": (_i64, [vp]),
    "egp_engine_k1_env_substeps": (_i64, [vp]),
    "egp_engine_event_overhead_ms": (C.c_double, [vp]),
    "egp_engine_set_profile": (C.c_int, [vp, C.c_int]),
    "egp_engine_layout": (C.c_int, [vp, c_int_p, c_int_p, c_int_p, c_int_p]),
    "egp_engine_group_range": (C.c_int, [vp, _i32, c_int_p, c_int_p]),
    "egp_engine_set_reward_job": (C.c_int, [vp, _i32, vp, vp, vp, vp, _f64, vp, vp]),
    "egp_engine_substeps_per_launch": (C.c_int, [vp]),
    "egp_engine_server_trace": (C.c_int, [vp, _i32, vp, vp]),
    "egp_engine_go_words_in_vram": (_i32, [vp]),
    "egp_engine_envs_per_wave": (_i32, [vp, C.POINTER(C.c_int32)]),
    "egp_host_probe": (C.c_int, [_i32, _i32, _i32, C.POINTER(HostProbeResult)]),
    "egp_device_usable_cus": (_i32, [_i32]),
}

_lib = None


class EgpError(RuntimeError):
    pass


def load():
    """dlopen the HIP library and type every entry point. Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EgpError(
            "egopose_amd: %s is missing -- build it with `python -m egopose_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise EgpError("egopose_amd: symbol %s not exported by %s" % (name, LIB_PATH)) from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    """C status -> Python exception (ValueError for bad arguments, RuntimeError otherwise)."""
    if rc == EGP_OK:
        return
    msg = load().egp_last_error().decode("utf-8", "replace")
    text = "%s failed (%d): %s" % (what or "egp call", rc, msg)
    if rc == -1:
        raise ValueError(text)
    raise EgpError(text)


def current_stream(device_index=None):
    """c_void_p of torch's current HIP stream on `device_index` (default: current device). Goes through the raw
    accessor (no Stream object is built); falls back to torch.cuda.current_stream."""
    import torch
    try:
        if device_index is None:
            device_index = torch._C._cuda_getDevice()
        return C.c_void_p(torch._C._cuda_getCurrentRawStream(device_index))
    except AttributeError:
        return C.c_void_p(torch.cuda.current_stream(device_index).cuda_stream)
