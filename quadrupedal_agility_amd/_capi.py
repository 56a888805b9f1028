"""ctypes mirror of include/qa_sim.h (struct qa_config, enums) and the loader of the HIP library.

The product path loads ONLY quadrupedal_agility_amd/csrc/libqa_sim.so (built by
__graft_entry__.build()).  It never falls back to a CPU path: a missing library raises.
"""
import ctypes as C
import os

QA_ABI_VERSION = 18
NUM_DOF = 12
NUM_GAITS = 5
NUM_PROP = 57
HISTORY_LEN = 10
NUM_OBS = 671
NUM_OBS_DISC = 49
ACTION_BUF_LEN = 8
NUM_REWARDS = 14
MOCAP_FRAME = 37
MAX_MOCAP_CLIPS, MOCAP_CLIP = 64, 8
OBST_SEESAW, OBST_BAR, OBST_TYRE = 1, 2, 3          # QA_OBST_* kinds of include/qa_sim.h

REWARD_NAMES = [
    "action_rate", "collision", "delta_torques", "dof_acc", "dof_error", "dof_pos_limits",
    "dof_vel_limits", "hip_pos", "jump_up_height", "locomotion_height", "torque_limits",
    "torques", "tracking_ang_vel", "tracking_lin_vel",
]

TENSORS = [
    "ROOT_STATES", "DOF_STATE", "CONTACT_FORCES", "RIGID_BODY_POS", "TORQUES", "TORQUES_ORG",
    "ACTIONS", "LAST_ACTIONS", "LAST_DOF_VEL", "LAST_TORQUES_ORG", "LAST_ROOT_VEL",
    "ACTION_HISTORY", "OBS", "OBS_DISC", "OBS_DISC_TERM", "COMMANDS",
    "LATENT_EPS", "LATENT_C", "REW", "RESET", "TIME_OUT", "EPISODE_LENGTH", "EPISODE_SUMS",
    "EPISODE_STATS", "LAST_CONTACTS", "CONTACT_FILT", "FEET_FORCE", "BASE_LIN_VEL",
    "BASE_ANG_VEL", "PROJECTED_GRAVITY", "RPY", "MOTOR_STRENGTH", "MASS_PARAMS", "FRICTION",
    "ENV_ORIGINS", "BASE_INERTIA", "PRIOR_PARAMETERS", "MOCAP_FRAMES", "HEIGHT_SAMPLES", "SCAN_HEIGHT", "FOOT_IMPULSE",
    "MOCAP_CLIPS", "RIGID_BODY_STATE", "STEP_TICKET", "CEILING_SAMPLES", "OBST_DESC", "OBST_STATE",
]
T = {name: i for i, name in enumerate(TENSORS)}
DTYPE_F32, DTYPE_I64, DTYPE_U8, DTYPE_I32, DTYPE_I16, DTYPE_F64 = 0, 1, 2, 3, 4, 5

BODY_NAMES = ["base", "Head_upper", "Head_lower"] + [
    f"{l}_{p}" for l in ("FL", "FR", "RL", "RR") for p in ("hip", "thigh", "calf", "foot")]
DOF_NAMES = [f"{l}_{p}_joint" for l in ("FL", "FR", "RL", "RR") for p in ("hip", "thigh", "calf")]


class QaConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("num_envs", C.c_int32), ("seed", C.c_uint64),
        ("sim_dt", C.c_float), ("decimation", C.c_int32), ("gravity_z", C.c_float),
        ("solver_iterations", C.c_int32), ("contact_offset", C.c_float),
        ("max_depenetration_velocity", C.c_float), ("ground_friction", C.c_float),
        ("terrain_type", C.c_int32),
        ("kp", C.c_float), ("kd", C.c_float), ("action_scale", C.c_float),
        ("hip_scale_reduction", C.c_float), ("clip_actions", C.c_float),
        ("default_dof_pos", C.c_float * NUM_DOF),
        ("env_spacing", C.c_float), ("max_episode_length", C.c_int32),
        ("resampling_steps", C.c_int32), ("push_interval", C.c_int32), ("push_robots", C.c_int32),
        ("max_push_vel_xy", C.c_float), ("reset_mode", C.c_int32), ("init_pos", C.c_float * 3),
        ("add_noise", C.c_int32),
        ("noise_roll_pitch", C.c_float), ("noise_ang_vel", C.c_float), ("noise_dof_pos", C.c_float),
        ("noise_dof_vel", C.c_float), ("noise_lin_vel", C.c_float), ("clip_obs", C.c_float),
        ("s_lin_vel", C.c_float), ("s_ang_vel", C.c_float), ("s_dof_pos", C.c_float),
        ("s_dof_vel", C.c_float), ("s_key_pos", C.c_float), ("s_foot_contact", C.c_float),
        ("s_lin_vel_dist", C.c_float), ("s_ang_vel_dist", C.c_float),
        ("reward_scale_dt", C.c_float * NUM_REWARDS), ("only_positive_rewards", C.c_int32),
        ("tracking_sigma", C.c_float), ("soft_dof_pos_limit", C.c_float),
        ("soft_dof_vel_limit", C.c_float), ("soft_torque_limit", C.c_float), ("jump_goal", C.c_float),
        ("lin_vel_x", (C.c_float * 2) * NUM_GAITS), ("lin_vel_y", (C.c_float * 2) * NUM_GAITS),
        ("ang_vel_yaw", (C.c_float * 2) * NUM_GAITS),
        ("jump_height", C.c_float * 2), ("locomotion_height", C.c_float * 2),
        ("lin_vel_x_clip", C.c_float), ("lin_vel_y_clip", C.c_float), ("ang_vel_yaw_clip", C.c_float),
        ("latent_temperature", C.c_float),
        ("randomize_friction", C.c_int32), ("randomize_base_mass", C.c_int32),
        ("randomize_base_com", C.c_int32), ("randomize_motor", C.c_int32), ("use_easi", C.c_int32),
        ("friction_range", C.c_float * 2), ("added_mass_range", C.c_float * 2),
        ("added_com_range", C.c_float * 2), ("motor_strength_range", C.c_float * 2),
        ("easi_mean", C.c_float * 6), ("easi_var", C.c_float * 6),
        ("hf_rows", C.c_int32), ("hf_cols", C.c_int32), ("hf_hscale", C.c_float), ("hf_vscale", C.c_float),
        ("hf_border", C.c_float), ("reset_xy_jitter", C.c_float),
        ("num_mocap_frames", C.c_int32), ("export_body_state", C.c_int32), ("env_id_offset", C.c_int32), ("num_envs_global", C.c_int32),
        ("contact_slots", C.c_int32), ("hf_ceiling", C.c_int32), ("articulated_obstacles", C.c_int32), ("self_collision", C.c_int32),
    ]


class QaDiscSampleIo(C.Structure):
    _fields_ = [("src", C.c_void_p * 3), ("index", C.c_void_p * 3), ("rows", C.c_int64 * 3), ("eps_src", C.c_void_p), ("c_src", C.c_void_p),
                ("eps_out", C.c_void_p), ("c_out", C.c_void_p), ("label_src", C.c_void_p), ("label_out", C.c_void_p), ("block_dev", C.c_void_p)]


def bind(lib, prefix):
    """Declare argtypes/restypes of the C ABI of include/qa_sim.h on a loaded library whose
    symbols carry `prefix` (the product library uses "qa_")."""
    P = C.POINTER
    f = getattr(lib, prefix + "arena_bytes"); f.argtypes = [P(QaConfig)]; f.restype = C.c_int64
    f = getattr(lib, prefix + "create"); f.argtypes = [P(QaConfig), C.c_void_p, C.c_int64, C.c_void_p, P(C.c_void_p)]; f.restype = C.c_int
    f = getattr(lib, prefix + "destroy"); f.argtypes = [C.c_void_p]; f.restype = C.c_int
    f = getattr(lib, prefix + "tensor_info"); f.argtypes = [P(QaConfig), C.c_int, P(C.c_int64), C.c_int64 * 3, P(C.c_int32), P(C.c_int32)]; f.restype = C.c_int
    f = getattr(lib, prefix + "env_step"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p]; f.restype = C.c_int
    f = getattr(lib, prefix + "env_step_dev"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]; f.restype = C.c_int
    f = getattr(lib, prefix + "reset_all"); f.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]; f.restype = C.c_int
    f = getattr(lib, prefix + "simulate"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]; f.restype = C.c_int
    f = getattr(lib, prefix + "set_mocap"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, P(C.c_int32), C.c_void_p]; f.restype = C.c_int
    f = getattr(lib, prefix + "debug_post_physics"); f.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]; f.restype = C.c_int
    f = getattr(lib, prefix + "env_physics_step"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]; f.restype = C.c_int
    f = getattr(lib, prefix + "tsc_reset"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_float] * 4 + [C.c_int64, C.c_void_p]; f.restype = C.c_int
    f = getattr(lib, prefix + "tsc_reset_dev"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_float] * 4 + [C.c_void_p, C.c_void_p]; f.restype = C.c_int
    f = getattr(lib, prefix + "simulate_if"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]; f.restype = C.c_int
    f = getattr(lib, prefix + "gae"); f.argtypes = [C.c_void_p] * 6 + [C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32, C.c_void_p, C.c_void_p]; f.restype = C.c_int
    f = getattr(lib, prefix + "ppo_loss")
    f.argtypes = [C.c_void_p] * 10 + [C.c_int64, C.c_int32] + [C.c_float] * 5 + [C.c_int32] + [C.c_void_p] * 5 + [C.c_int64, C.c_void_p]
    f.restype = C.c_int
    f = getattr(lib, prefix + "ppo_loss_scratch_bytes"); f.argtypes = [C.c_int64]; f.restype = C.c_int64
    f = getattr(lib, prefix + "hybrid_ppo_loss_scratch_bytes"); f.argtypes = [C.c_int64]; f.restype = C.c_int64
    f = getattr(lib, prefix + "hybrid_ppo_loss")
    f.argtypes = [C.c_void_p] * 12 + [C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_int32] + [C.c_void_p] * 6 + [C.c_int64, C.c_void_p]
    f.restype = C.c_int
    f = getattr(lib, prefix + "elu_backward_bias_scratch_bytes"); f.argtypes = [C.c_int64, C.c_int32]; f.restype = C.c_int64
    f = getattr(lib, prefix + "normalizer_update")
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    f.restype = C.c_int
    f = getattr(lib, prefix + "normalizer_apply")
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
    f.restype = C.c_int
    f = getattr(lib, prefix + "clip_adam_step")
    f.argtypes = [C.c_void_p] * 5 + [C.c_int32] + [C.c_void_p] * 3 + [C.c_int32, C.c_void_p, C.c_void_p] + [C.c_float] * 4 + [C.c_void_p, C.c_int64, C.c_void_p]
    f.restype = C.c_int
    f = getattr(lib, prefix + "clip_adam_step_hostgrads")
    f.argtypes = [C.c_void_p] * 5 + [C.c_int32] + [C.c_void_p] * 3 + [C.c_int32, C.c_void_p, C.c_void_p] + [C.c_float] * 4 + [C.c_void_p, C.c_int64, C.c_void_p]
    f.restype = C.c_int
    f = getattr(lib, prefix + "clip_adam_step_reduce")
    f.argtypes = [C.c_void_p] * 5 + [C.c_int32] + [C.c_void_p] * 3 + [C.c_int32, C.c_void_p, C.c_void_p] + [C.c_float] * 4 + [C.c_void_p, C.c_int64] + [C.c_void_p] * 4
    f.restype = C.c_int
    f = getattr(lib, prefix + "grad_reduce"); f.argtypes = [C.c_void_p] * 5 + [C.c_int32, C.c_void_p]; f.restype = C.c_int
    f = getattr(lib, prefix + "clip_adam_pair_step")
    f.argtypes = [C.c_void_p] * 5 + [C.c_int32] + [C.c_void_p] * 3 + [C.c_int32, C.c_void_p, C.c_void_p] + [C.c_float] * 4 + [C.c_void_p, C.c_int64] + [C.c_void_p] * 5
    f.restype = C.c_int
    f = getattr(lib, prefix + "pair_losses_scratch_bytes"); f.argtypes = [C.c_void_p, C.c_int32]; f.restype = C.c_int64
    f = getattr(lib, prefix + "pair_losses"); f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]; f.restype = C.c_int
    f = getattr(lib, prefix + "accumulate_scalars"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]; f.restype = C.c_int
    f = getattr(lib, prefix + "adam_stack_step"); f.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]; f.restype = C.c_int
    f = getattr(lib, prefix + "rollout_act")
    f.argtypes = [C.c_void_p] * 4 + [C.c_uint64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32] + [C.c_void_p] * 7
    f.restype = C.c_int
    f = getattr(lib, prefix + "rollout_act_store")
    f.argtypes = [C.c_void_p] * 4 + [C.c_uint64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32] + [C.c_void_p] * 6 + [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]
    f.restype = C.c_int
    f = getattr(lib, prefix + "rollout_post")
    f.argtypes = [C.c_void_p] * 4 + [C.c_float, C.c_float, C.c_int32] + [C.c_void_p] * 6
    f.restype = C.c_int
    f = getattr(lib, prefix + "rollout_post_amp")
    f.argtypes = ([C.c_void_p] * 7 + [C.c_int32, C.c_void_p, C.c_int64, C.c_int32] + [C.c_float] * 6 + [C.c_int32] + [C.c_void_p] * 6)
    f.restype = C.c_int
    f = getattr(lib, prefix + "disc_loss_scratch_bytes"); f.argtypes = [C.c_int64]; f.restype = C.c_int64
    f = getattr(lib, prefix + "disc_loss")
    f.argtypes = [C.c_void_p] * 6 + [C.c_int32] * 3 + [C.c_float, C.c_void_p, C.c_float, C.c_float] + [C.c_void_p] * 5 + [C.c_int64, C.c_void_p]
    f.restype = C.c_int
    f = getattr(lib, prefix + "disc_loss_logits")
    f.argtypes = [C.c_void_p] * 6 + [C.c_int32] * 3 + [C.c_float, C.c_void_p, C.c_float, C.c_float] + [C.c_void_p] * 5 + [C.c_int64, C.c_void_p]
    f.restype = C.c_int
    f = getattr(lib, prefix + "disc_prepare")
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 5 + [C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    f.restype = C.c_int
    f = getattr(lib, prefix + "pair_loss_scratch_bytes"); f.argtypes = [C.c_int64]; f.restype = C.c_int64
    f = getattr(lib, prefix + "pair_loss")
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    f.restype = C.c_int
    f = getattr(lib, prefix + "gather_rows"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32] + [C.c_void_p] * 5; f.restype = C.c_int
    f = getattr(lib, prefix + "kl_lr_rule"); f.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]; f.restype = C.c_int
    f = getattr(lib, prefix + "tsc_push"); f.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_uint64, C.c_int32, C.c_void_p]; f.restype = C.c_int
    f = getattr(lib, prefix + "tsc_start_pose")
    f.argtypes = [C.c_void_p] * 4 + [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_uint64, C.c_void_p, C.c_int32] + [C.c_void_p] * 4
    f.restype = C.c_int
    f = getattr(lib, prefix + "tsc_reset_where")
    f.argtypes = [C.c_void_p] * 6 + [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    f.restype = C.c_int
    f = getattr(lib, prefix + "rollout_act_hybrid")
    f.argtypes = [C.c_void_p] * 4 + [C.c_uint64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 9 + [C.c_int32, C.c_void_p]
    f.restype = C.c_int
    f = getattr(lib, prefix + "set_lean_exports"); f.argtypes = [C.c_void_p, C.c_int32]; f.restype = C.c_int
    f = getattr(lib, prefix + "episode_means"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]; f.restype = C.c_int
    f = getattr(lib, prefix + "tsc_set_commands")
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 10
    f.restype = C.c_int
    f = getattr(lib, prefix + "tsc_goal_step"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]; f.restype = C.c_int
    f = getattr(lib, prefix + "tsc_observations"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]; f.restype = C.c_int
    f = getattr(lib, prefix + "tsc_depth_update"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]; f.restype = C.c_int
    f = getattr(lib, prefix + "mlp_packed_floats"); f.argtypes = [C.c_void_p, C.c_int32]; f.restype = C.c_int64
    f = getattr(lib, prefix + "mlp_pack"); f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]; f.restype = C.c_int
    f = getattr(lib, prefix + "mlp_strands"); f.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]; f.restype = C.c_int
    f = getattr(lib, prefix + "mlp_groups"); f.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]; f.restype = C.c_int
    f = getattr(lib, prefix + "mlp_set_groups"); f.argtypes = [C.c_int32]; f.restype = C.c_int
    f = getattr(lib, prefix + "mlp_forward")
    f.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    f.restype = C.c_int
    f = getattr(lib, prefix + "elu_backward_bias")
    f.argtypes = [C.c_void_p] * 4 + [C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_int64, C.c_void_p]
    f.restype = C.c_int
    f = getattr(lib, prefix + "narrow_wgrad_scratch_bytes"); f.argtypes = [C.c_int64, C.c_int32, C.c_int32]; f.restype = C.c_int64
    f = getattr(lib, prefix + "linear_forward")
    f.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p]
    f.restype = C.c_int
    f = getattr(lib, prefix + "linear_backward_input")
    f.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p]
    f.restype = C.c_int
    f = getattr(lib, prefix + "linear_backward_weight_scratch_bytes"); f.argtypes = [C.c_int64, C.c_int32, C.c_int32]; f.restype = C.c_int64
    f = getattr(lib, prefix + "linear_backward_weight_layout"); f.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_void_p]; f.restype = C.c_int
    f = getattr(lib, prefix + "linear_backward_weight_batch_layout"); f.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_void_p]; f.restype = C.c_int
    f = getattr(lib, prefix + "linear_backward_weight_batch_scratch_bytes"); f.argtypes = [C.c_int64, C.c_int32, C.c_int32]; f.restype = C.c_int64
    f = getattr(lib, prefix + "linear_backward_weight_batch"); f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]; f.restype = C.c_int
    f = getattr(lib, prefix + "linear_backward_weight")
    f.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]
    f.restype = C.c_int
    f = getattr(lib, prefix + "tsc_reset_stats"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]; f.restype = C.c_int
    f = getattr(lib, prefix + "slab_sum"); f.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]; f.restype = C.c_int
    P, I64, I32, FL = C.c_void_p, C.c_int64, C.c_int32, C.c_float
    f = getattr(lib, prefix + "linear_forward_split_scratch_bytes"); f.argtypes = [I64, I32, I32]; f.restype = C.c_int64
    f = getattr(lib, prefix + "linear_forward_split"); f.argtypes = [P, I64, P, I64, P, P, I64, I64, I32, I32, I32, FL, P, I64, P]; f.restype = C.c_int
    f = getattr(lib, prefix + "disc_sample_prepare"); f.argtypes = [C.POINTER(QaDiscSampleIo), I32, I32, P, P, P, P, P, FL, FL, P, P]; f.restype = C.c_int
    f = getattr(lib, prefix + "disc_step_tail_scratch_bytes"); f.argtypes = []; f.restype = C.c_int64
    f = getattr(lib, prefix + "disc_step_tail"); f.argtypes = [P, P, I64, I32, P, P, I32, P, P, P, P, I32, FL, P, I64, P]; f.restype = C.c_int
    f = getattr(lib, prefix + "depth_stem_forward"); f.argtypes = [P, P, P, P, P, I64, I32, I32, FL, P]; f.restype = C.c_int
    f = getattr(lib, prefix + "depth_stem_backward_scratch_bytes"); f.argtypes = []; f.restype = C.c_int64
    f = getattr(lib, prefix + "depth_stem_backward"); f.argtypes = [P, P, P, P, I64, I32, I32, P, I64, P]; f.restype = C.c_int
    f = getattr(lib, prefix + "conv_nhwc_forward"); f.argtypes = [P, P, P, P, I64, I32, I32, I32, I32, I32, I32, I32, FL, P]; f.restype = C.c_int
    f = getattr(lib, prefix + "conv_nhwc_backward_input"); f.argtypes = [P, P, P, P, I64, I32, I32, I32, I32, I32, I32, I32, FL, P]; f.restype = C.c_int
    f = getattr(lib, prefix + "conv_nhwc_backward_weight_scratch_bytes"); f.argtypes = [I64, I32, I32, I32, I32, I32, I32]; f.restype = C.c_int64
    f = getattr(lib, prefix + "conv_nhwc_backward_weight"); f.argtypes = [P, P, P, P, I64, I32, I32, I32, I32, I32, I32, P, I64, P]; f.restype = C.c_int
    f = getattr(lib, prefix + "elu_backward_pad"); f.argtypes = [P, P, P, P, I64, I32, I32, I32, I32, I32, FL, P]; f.restype = C.c_int
    f = getattr(lib, prefix + "narrow_wgrad"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]; f.restype = C.c_int
    return lib


class QaMlpOp(C.Structure):
    """qa_mlp_op of include/qa_sim.h"""
    _fields_ = [("kind", C.c_int32), ("src_buf", C.c_int32), ("src_col", C.c_int32), ("dst_buf", C.c_int32), ("dst_col", C.c_int32),
                ("k", C.c_int32), ("n", C.c_int32), ("act", C.c_int32), ("out_index", C.c_int32), ("flags", C.c_int32),
                ("w_off", C.c_int64), ("b_off", C.c_int64), ("out_col", C.c_int32), ("aux_index", C.c_int32), ("aux_col", C.c_int32), ("pad_", C.c_int32)]


class QaWgradDesc(C.Structure):
    """qa_wgrad_desc of include/qa_sim.h"""
    _fields_ = [("grad_out", C.c_void_p), ("ldg", C.c_int64), ("x", C.c_void_p), ("ldx", C.c_int64), ("grad_weight", C.c_void_p), ("grad_bias", C.c_void_p),
                ("rows", C.c_int64), ("in_features", C.c_int32), ("out_features", C.c_int32), ("scratch", C.c_void_p), ("scratch_bytes", C.c_int64)]


ADAM_STACK_MAX_TENSORS, ADAM_STACK_MAX_STATES = 16, 3


class QaPairJob(C.Structure):
    """qa_pair_job of include/qa_sim.h"""
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("rows", C.c_int64), ("cols", C.c_int32), ("mode", C.c_int32), ("b_stride", C.c_int64),
                ("grad_scale", C.c_void_p), ("grad_a", C.c_void_p), ("out", C.c_void_p)]


class QaAdamPair(C.Structure):
    """qa_adam_pair of include/qa_sim.h"""
    _fields_ = [("split_tensor", C.c_int32), ("split_chunk", C.c_int32), ("lr2", C.c_void_p), ("max_norm2", C.c_float), ("kl", C.c_void_p),
                ("desired_kl", C.c_float), ("kl_factor", C.c_float), ("lr_min", C.c_float), ("lr_max", C.c_float)]


class QaAdamStackState(C.Structure):
    """qa_adam_stack_state of include/qa_sim.h"""
    _fields_ = [("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("step", C.c_void_p), ("lr", C.c_void_p), ("weight_decay", C.c_float), ("pad_", C.c_int32)]


class QaAdamStackTensor(C.Structure):
    """qa_adam_stack_tensor of include/qa_sim.h"""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("tmp", C.c_void_p), ("src1", C.c_void_p), ("src2", C.c_void_p), ("stride1", C.c_int64), ("stride2", C.c_int64),
                ("parts1", C.c_int32), ("parts2", C.c_int32), ("alpha2", C.c_float), ("reg", C.c_float), ("numel", C.c_int32), ("num_states", C.c_int32),
                ("state", QaAdamStackState * ADAM_STACK_MAX_STATES)]


MLP_COPY, MLP_LAYER, MLP_GRAD, MLP_LOAD, MLP_MAX_OPS, MLP_MAX_OUTPUTS = 0, 1, 2, 3, 24, 8
MLP_F_SAVE, MLP_F_TRANSPOSED, MLP_F_ADD = 1, 2, 4
MLP_ACT_ELU_GRAD, MLP_ACT_RELU_GRAD, MLP_ACT_TANH_GRAD = 4, 5, 6
MLP_BUF_COLS = (800, 576, 320, 128)

TSC_REWARD_NAMES = ("action_hl_rate", "collision", "feet_edge", "latent_c_rate", "reach_goal", "tracking_goal_vel", "tracking_yaw",
                    "termination")          # QA_TSC_REW_* order
TSC_MAX_BODY_IDS = 24


class QaTscGoalCfg(C.Structure):
    """qa_tsc_goal_cfg of include/qa_sim.h"""
    _fields_ = [("num_envs", C.c_int64), ("num_bodies", C.c_int32), ("num_goal_slots", C.c_int32), ("last_goal_repeat", C.c_int32),
                ("goals_per_obstacle", C.c_int32), ("num_obstacles", C.c_int32), ("history_len", C.c_int32), ("history_width", C.c_int32),
                ("mask_rows", C.c_int32), ("mask_cols", C.c_int32), ("use_camera", C.c_int32),
                ("num_termination_bodies", C.c_int32), ("num_penalised_bodies", C.c_int32),
                ("termination_bodies", C.c_int32 * TSC_MAX_BODY_IDS), ("penalised_bodies", C.c_int32 * TSC_MAX_BODY_IDS),
                ("feet_bodies", C.c_int32 * 4), ("reach_goal_delay_steps", C.c_float), ("next_goal_threshold", C.c_float),
                ("leave_goal_threshold", C.c_float), ("max_episode_length", C.c_float), ("target_lin_vel", C.c_float),
                ("border_size", C.c_float), ("horizontal_scale", C.c_float), ("reward_scales", C.c_float * len(TSC_REWARD_NAMES))]


TSC_GOAL_IO_FIELDS = ("root_states", "contact_forces", "rigid_body_states", "env_goals", "obstacle_types", "action_hl_history", "x_edge_mask",
                      "episode_length", "cur_goal_idx", "reach_goal_timer", "last_contacts", "cur_goals", "next_goals", "episode_sums",
                      "base_lin_vel", "base_ang_vel", "projected_gravity", "rpy", "contact_filt", "target_pos_rel", "next_target_pos_rel",
                      "target_yaw", "next_target_yaw", "reached_goal", "cur_obstacle_type", "reset_buf", "time_out_buf", "reach_goal_cutoff",
                      "rew_buf")


class QaTscGoalIo(C.Structure):
    """qa_tsc_goal_io of include/qa_sim.h (every member is a pointer)"""
    _fields_ = [(name, C.c_void_p) for name in TSC_GOAL_IO_FIELDS]


class QaTscObsCfg(C.Structure):
    """qa_tsc_obs_cfg of include/qa_sim.h"""
    _fields_ = [("num_envs", C.c_int64), ("num_bodies", C.c_int32), ("key_bodies", C.c_int32 * 4), ("map_rows", C.c_int32), ("map_cols", C.c_int32),
                ("update_yaw", C.c_int32), ("root_height_obs", C.c_int32), ("action_stride", C.c_int64),
                ("points_env_stride", C.c_int64), ("point_stride", C.c_int32), ("reserved", C.c_int32),
                ("border_size", C.c_float), ("horizontal_scale", C.c_float), ("vertical_scale", C.c_float),
                ("lin_vel", C.c_float), ("ang_vel", C.c_float), ("dof_pos", C.c_float), ("dof_vel", C.c_float), ("lin_vel_dist", C.c_float),
                ("ang_vel_dist", C.c_float), ("key_pos", C.c_float), ("foot_contact", C.c_float), ("clip_observations", C.c_float),
                ("default_dof_pos", C.c_float * 12), ("default_dof_pos_all", C.c_float * 12)]


TSC_OBS_IO_FIELDS = ("root_states", "rpy", "base_lin_vel", "base_ang_vel", "contact_filt", "dof_pos", "dof_vel", "last_action",
                     "rigid_body_states", "mass_params", "friction", "motor_strength", "cur_obstacle_type", "target_yaw", "next_target_yaw",
                     "height_samples", "height_points", "commands", "latent_eps", "latent_c", "episode_length", "delta_yaw", "delta_next_yaw",
                     "obs_history", "measured_heights", "obs_buf", "obs_bbc_buf", "obs_disc_buf")
TSC_NUM_SCAN, TSC_NUM_OBS, TSC_NUM_OBS_BBC, TSC_NUM_OBS_DISC = 132, 800, 671, 49


class QaTscObsIo(C.Structure):
    """qa_tsc_obs_io of include/qa_sim.h (every member is a pointer)"""
    _fields_ = [(name, C.c_void_p) for name in TSC_OBS_IO_FIELDS]


class QaTscDepthCfg(C.Structure):
    """qa_tsc_depth_cfg of include/qa_sim.h"""
    _fields_ = [("num_envs", C.c_int64), ("step", C.c_int64), ("seed", C.c_uint64), ("env_id_offset", C.c_int32), ("width", C.c_int32),
                ("height", C.c_int32), ("crop_top", C.c_int32), ("crop_bottom", C.c_int32), ("crop_left", C.c_int32), ("crop_right", C.c_int32),
                ("buffer_len", C.c_int32), ("map_rows", C.c_int32), ("map_cols", C.c_int32), ("coarse_log2", C.c_int32),
                ("horizontal_fov_deg", C.c_float), ("position", C.c_float * 3), ("near_clip", C.c_float), ("far_clip", C.c_float),
                ("depth_noise", C.c_float), ("border_size", C.c_float), ("horizontal_scale", C.c_float), ("vertical_scale", C.c_float)]


TSC_DEPTH_IO_FIELDS = ("root_states", "camera_pitch", "height_samples", "ceiling_samples", "episode_length", "depth_buffer", "coarse_floor_max",
                       "coarse_ceiling_min", "step_dev")


class QaTscDepthIo(C.Structure):
    """qa_tsc_depth_io of include/qa_sim.h (every member is a pointer)"""
    _fields_ = [(name, C.c_void_p) for name in TSC_DEPTH_IO_FIELDS]


ABI_SYMBOLS = ["arena_bytes", "create", "destroy", "tensor_info", "env_step", "env_step_dev", "reset_all", "simulate",
               "set_mocap", "debug_post_physics", "env_physics_step", "tsc_reset", "tsc_reset_dev", "simulate_if", "gae", "ppo_loss", "ppo_loss_scratch_bytes", "hybrid_ppo_loss", "hybrid_ppo_loss_scratch_bytes", "elu_backward_bias",
               "elu_backward_bias_scratch_bytes", "narrow_wgrad", "narrow_wgrad_scratch_bytes", "linear_forward", "linear_backward_input", "linear_backward_weight", "linear_backward_weight_scratch_bytes", "linear_backward_weight_layout", "linear_backward_weight_batch", "linear_backward_weight_batch_layout", "linear_backward_weight_batch_scratch_bytes", "slab_sum", "linear_forward_split", "linear_forward_split_scratch_bytes", "depth_stem_forward", "depth_stem_backward", "depth_stem_backward_scratch_bytes", "conv_nhwc_forward", "conv_nhwc_backward_input", "conv_nhwc_backward_weight", "conv_nhwc_backward_weight_scratch_bytes", "elu_backward_pad", "normalizer_update", "normalizer_apply", "clip_adam_step", "clip_adam_step_hostgrads", "clip_adam_step_reduce", "clip_adam_pair_step", "grad_reduce", "adam_stack_step", "rollout_act", "rollout_act_store", "rollout_act_hybrid", "rollout_post", "rollout_post_amp", "disc_loss", "disc_loss_logits", "disc_loss_scratch_bytes", "disc_prepare", "disc_sample_prepare", "disc_step_tail", "disc_step_tail_scratch_bytes", "pair_loss", "pair_loss_scratch_bytes", "pair_losses", "pair_losses_scratch_bytes", "accumulate_scalars", "gather_rows", "kl_lr_rule", "episode_means", "set_lean_exports", "mlp_packed_floats", "mlp_pack", "mlp_forward", "mlp_strands", "mlp_groups", "mlp_set_groups", "tsc_set_commands", "tsc_goal_step", "tsc_observations", "tsc_depth_update", "tsc_reset_stats", "tsc_push", "tsc_start_pose", "tsc_reset_where", "last_error", "abi_version"]

_LIB = None
# QA_LIB: another build of the SAME library (A/B measurements of a kernel variant, tools/r5_call.sh); there is still no fallback -- a missing file raises
LIB_PATH = os.environ.get("QA_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libqa_sim.so")


def load_library():
    """Load the HIP library; raise loudly if it has not been built (no CPU fallback exists)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). quadrupedal_agility_amd has no CPU fallback.")
        # PyTorch ships its own HIP runtime (torch/lib/libamdhip64.so); the device pointers and streams this library is
        # handed belong to THAT runtime.  Loading torch first makes our DT_NEEDED libamdhip64 resolve to the copy already in
        # the process -- loaded the other way round, the process holds two runtimes and ours sees no device.
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        bind(lib, "qa_")
        lib.qa_last_error.restype = C.c_char_p
        lib.qa_abi_version.restype = C.c_int
        if lib.qa_abi_version() != QA_ABI_VERSION:
            raise RuntimeError("libqa_sim.so ABI version mismatch")
        _LIB = lib
    return _LIB


def tensor_info(lib, prefix, cfg, which):
    off = C.c_int64(); shape = (C.c_int64 * 3)(); nd = C.c_int32(); dt = C.c_int32()
    rc = getattr(lib, prefix + "tensor_info")(C.byref(cfg), which, C.byref(off), shape, C.byref(nd), C.byref(dt))
    if rc != 0:
        raise RuntimeError(f"{prefix}tensor_info({which}) -> {rc}")
    return off.value, tuple(shape[i] for i in range(nd.value)), dt.value
