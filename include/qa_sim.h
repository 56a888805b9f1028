/* qa_sim.h -- C ABI of the MI355X-native Go2 vectorised environment ("quadrupedal-agility_amd").
 *
 * This is the drop-in boundary for the ONE hot path this build replaces: the
 * legged_gym `LeggedRobot.step()` loop of NJU-RLC/quadrupedal-agility, i.e. everything the
 * reference does between receiving `actions (N,12)` and returning obs/reward/reset
 * (bbc/legged_gym/envs/base/legged_robot.py:78-115), including the calls it makes into the
 * closed-source Isaac Gym tensor API (seam 1), plus the GAE scan of
 * the learner (bbc/rsl_rl/storage/rollout_storage.py:97-111).
 *
 * Conventions (same as the reference's gym tensor API, legged_robot.py:747-770):
 *   - all state lives in ONE caller-allocated device slab ("arena"); the engine lays the
 *     tensors out inside it and the caller aliases them zero-copy (torch views).  The env
 *     both reads and writes those aliases in place, exactly like gymtorch.wrap_tensor views.
 *   - row-major, fp32 unless stated; quaternions xyzw; root velocities in the world frame;
 *     DoF order FL,FR,RL,RR x (hip, thigh, calf); env ids are int32.
 *   - every entry point returns 0 on success or a negative QA_E_* code; nothing aborts.
 *   - single host thread per handle; device work is enqueued on the stream passed in
 *     (hipStream_t as void*; NULL = default stream); no host-visible sync is performed.
 *
 * The CPU oracle (oracle/qa_oracle.c) implements the same functions with the prefix `qo_`
 * over a host arena with the identical layout, so tests can memcpy an arena across.
 */
#ifndef QA_SIM_H
#define QA_SIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QA_ABI_VERSION 18
#define QA_NUM_DOF 12
#define QA_NUM_BODIES_ABI 19
#define QA_NUM_GAITS 5          /* walk, pace, trot, canter, jump (go2_locomotion_config.py:24) */
#define QA_NUM_PROP 57
#define QA_HISTORY_LEN 10
#define QA_NUM_OBS 671          /* 57 + 4 + 29 + 570 + 11 (legged_robot.py:261-331) */
#define QA_NUM_OBS_DISC 49
#define QA_ACTION_BUF_LEN 8
#define QA_NUM_REWARDS 14

/* error codes */
#define QA_OK 0
#define QA_E_ARG (-1)           /* null / out-of-range argument */
#define QA_E_ARENA (-2)         /* arena too small or misaligned */
#define QA_E_DEVICE (-3)        /* HIP runtime error (qa_last_error() has the text) */
#define QA_E_VERSION (-4)

/* reward terms, alphabetical = the order the reference sums them in
 * (helpers.py:12-27 class_to_dict is dir()-sorted; legged_robot.py:922-946) */
enum qa_reward {
    QA_R_ACTION_RATE = 0, QA_R_COLLISION, QA_R_DELTA_TORQUES, QA_R_DOF_ACC, QA_R_DOF_ERROR,
    QA_R_DOF_POS_LIMITS, QA_R_DOF_VEL_LIMITS, QA_R_HIP_POS, QA_R_JUMP_UP_HEIGHT,
    QA_R_LOCOMOTION_HEIGHT, QA_R_TORQUE_LIMITS, QA_R_TORQUES, QA_R_TRACKING_ANG_VEL,
    QA_R_TRACKING_LIN_VEL
};

/* tensors inside the arena.  Layout rule (part of the ABI): tensors are placed in enum
 * order, each start rounded up to 256 bytes. */
enum qa_tensor {
    QA_T_ROOT_STATES = 0,     /* (N,13)  pos3 quat4(xyzw) linvel3 angvel3, world frame          */
    QA_T_DOF_STATE,           /* (N,12,2) [pos, vel] interleaved like gym's dof_state            */
    QA_T_CONTACT_FORCES,      /* (N,19,3) net contact force per body, world frame, last substep  */
    QA_T_RIGID_BODY_POS,      /* (N,19,3) body-origin positions, world frame (refreshed by step) */
    QA_T_TORQUES,             /* (N,12)  clipped torques of the last substep                     */
    QA_T_TORQUES_ORG,         /* (N,12)  un-clipped torques of the last substep                  */
    QA_T_ACTIONS,             /* (N,12)  clipped (and possibly delayed) actions                  */
    QA_T_LAST_ACTIONS,        /* (N,12)                                                          */
    QA_T_LAST_DOF_VEL,        /* (N,12)                                                          */
    QA_T_LAST_TORQUES_ORG,    /* (N,12)                                                          */
    QA_T_LAST_ROOT_VEL,       /* (N,6)                                                           */
    QA_T_ACTION_HISTORY,      /* (N,8,12) oldest first                                           */
    QA_T_OBS,                 /* (N,671) obs_buf == privileged_obs_buf.  Columns [90:660] ARE the proprioceptive
                                         history (10 x 57, oldest first, noise-free): the reference's separate
                                         obs_history_buf is this slice, kept in place from step to step     */
    QA_T_OBS_DISC,            /* (N,49)  discriminator observation                               */
    QA_T_OBS_DISC_TERM,       /* (N,49)  OBS_DISC with rows of resetting envs replaced by their
                                         terminal (pre-reset) disc obs                           */
    QA_T_COMMANDS,            /* (N,5)   vx vy wz jump_height locomotion_height                  */
    QA_T_LATENT_EPS,          /* (N,1)                                                           */
    QA_T_LATENT_C,            /* (N,5)   one-hot gait                                            */
    QA_T_REW,                 /* (N)                                                             */
    QA_T_RESET,               /* (N) int64                                                       */
    QA_T_TIME_OUT,            /* (N) uint8                                                       */
    QA_T_EPISODE_LENGTH,      /* (N) int64                                                       */
    QA_T_EPISODE_SUMS,        /* (14,N)                                                          */
    QA_T_EPISODE_STATS,       /* (2,16)  [parity][0:14]=sum over resetting envs of episode sums,
                                         [14]=number of resetting envs; parity = global_step & 1.
                                         A step accumulates into its own bin and clears the bin
                                         of the next step, so global_step must advance by 1.     */
    QA_T_LAST_CONTACTS,       /* (N,4) uint8                                                     */
    QA_T_CONTACT_FILT,        /* (N,4) uint8                                                     */
    QA_T_FEET_FORCE,          /* (N,4)  norm of foot contact forces                              */
    QA_T_BASE_LIN_VEL,        /* (N,3)  body frame                                               */
    QA_T_BASE_ANG_VEL,        /* (N,3)  body frame                                               */
    QA_T_PROJECTED_GRAVITY,   /* (N,3)                                                           */
    QA_T_RPY,                 /* (N,3)  roll pitch yaw                                           */
    QA_T_MOTOR_STRENGTH,      /* (2,N,12) p and d multipliers                                    */
    QA_T_MASS_PARAMS,         /* (N,4)  added base mass, added CoM xyz                           */
    QA_T_FRICTION,            /* (N)    robot-shape friction coefficient                         */
    QA_T_ENV_ORIGINS,         /* (N,3)                                                           */
    QA_T_BASE_INERTIA,        /* (N,10) base link incl. added mass: m, m*c (3), I about base
                                         origin xx yy zz xy xz yz, base frame                    */
    QA_T_PRIOR_PARAMETERS,    /* (5)    written by the learner (gail.py:463-464)                 */
    QA_T_MOCAP_FRAMES,        /* (F,QA_MOCAP_FRAME) frames of the labelled mocap clips, see qa_set_mocap */
    QA_T_HEIGHT_SAMPLES,      /* (hf_rows, hf_cols) int16 terrain height samples (terrain_type 1), written by the caller
                                         after qa_create exactly like gym.add_heightfield's samples
                                         (legged_robot.py:958-975): height = sample * hf_vscale at
                                         x = row * hf_hscale - hf_border, y = col * hf_hscale - hf_border        */
    QA_T_SCAN_HEIGHT,         /* (N) terrain height under scan point 94 of the reference's 17x11 height scan, the only
                                         sample the BBC env uses (root_h = z - measured_heights[:, 94], :264-268); 0 on a plane */
    QA_T_FOOT_IMPULSE,        /* (N,4,3) foot contact impulses (normal, tangent x, tangent y) of the last
                                         substep: warm start of the contact solver; zeroed on reset   */
    QA_T_MOCAP_CLIPS,         /* (QA_MAX_MOCAP_CLIPS,QA_MOCAP_CLIP) float64 clip table of qa_set_mocap            */
    QA_T_RIGID_BODY_STATE,    /* (N,19,13) pos3 quat4(xyzw) linvel3 angvel3 per body, world frame -- the reference's
                                         rigid_body_state viewed as (N, num_bodies, 13) (legged_robot.py:759-768).
                                         Written by step / simulate only when cfg.export_body_state != 0 (else one row) */
    QA_T_STEP_TICKET,         /* (4) int32: arrival counter of qa_env_step_dev's last-workgroup step-counter update */
    QA_T_CEILING_SAMPLES,     /* (hf_rows, hf_cols) int16, only with cfg.hf_ceiling (else one element): height of the UNDERSIDE of an
                                         overhang above the cell (tunnel roof, upper arc of the tyre), same grid and scales as
                                         HEIGHT_SAMPLES; QA_NO_CEILING where there is none.  A contact candidate collides with the nearer of
                                         floor and ceiling; the ceiling's contact normal points down (away from the obstacle)      */
    QA_T_OBST_DESC,           /* (N,3,8) fp32, only with cfg.articulated_obstacles (else one row): the env's three articulated course
                                         obstacles -- slot 0 see-saw, 1 bar jump, 2 tyre jump -- as
                                         [origin x, origin y (world), cos yaw, sin yaw, half length (along the obstacle's x), half width,
                                          pivot height h0 (see-saw), kind (0 none, QA_OBST_SEESAW, QA_OBST_BAR, QA_OBST_TYRE)]: the moving
                                         part's footprint in the obstacle frame (tsc/legged_gym/envs/base/legged_robot.py:1411-1427)      */
    QA_T_OBST_STATE,          /* (N,3,4) fp32: [q, q_dot, generalised contact force accumulated over the substeps of the running env
                                         step, joint damping (see-saw: U(1,10) N m s/rad)]: the obstacle joints' dof state
                                         (obst_dof_pos / obst_dof_vel of the reference, :792-794, :812-823): see-saw tilt [rad] about
                                         the obstacle's y axis, bar / tyre vertical offset [m] from the height the course map draws them at */
    QA_T_COUNT
};

enum qa_dtype { QA_F32 = 0, QA_I64 = 1, QA_U8 = 2, QA_I32 = 3, QA_I16 = 4, QA_F64 = 5 };

/* articulated course obstacles (QA_T_OBST_DESC / QA_T_OBST_STATE).  Inertias and the position drive are the reference's:
 * the link0 inertials of tsc/resources/obstacles/{seesaw,bar_jump,tire_jump} (URDF), legged_robot.py:1411-1427 (stiffness 20000, damping 1000; see-saw: no
 * stiffness, damping drawn per env), joint velocity limit 8 rad/s, see-saw travel +-asin(0.25 / 1.5) (utils/obstacle.py seesaw_dof_pos). */
#define QA_OBST_PER_ENV 3
#define QA_OBST_DESC 8
#define QA_OBST_STATE 4
#define QA_OBST_SEESAW 1
#define QA_OBST_BAR 2
#define QA_OBST_TYRE 3
#define QA_SEESAW_INERTIA 10.833f
#define QA_SEESAW_MAX_TILT 0.16744808f
#define QA_SEESAW_MAX_VEL 8.0f
#define QA_SEESAW_SHELL 0.10f    /* m: the plank's top stops points down to this far below it (its underside is not modelled) */
#define QA_BAR_MASS 3.39292f
#define QA_TYRE_MASS 13.038f
#define QA_OBST_STIFFNESS 20000.0f
#define QA_OBST_DAMPING 1000.0f

#define QA_MOCAP_FRAME 37       /* root pos3, root quat4, joint pos12, lin vel3, ang vel3 (root frame), joint vel12 */
#define QA_CEILING_SHELL 0.05f   /* m: an overhang's underside stops points up to this far above it (thin-shell rule) */
#define QA_NO_CEILING 32767
#define QA_MAX_MOCAP_CLIPS 64
#define QA_MOCAP_CLIP 8         /* first frame row, number of frames n, clip length (n-1) frame_duration [s], sampling range
                                   length - (time_between_frames disc_obs_len + frame_duration) [s], cumulative sampling
                                   probability inside the clip's gait (the last clip of a gait has 1), 3 unused */

/* Plain-old-data configuration.  Field meanings follow the reference config classes
 * (bbc/legged_gym/envs/go2/go2_locomotion_config.py, envs/base/legged_robot_config.py). */
typedef struct qa_config {
    int32_t abi_version;            /* QA_ABI_VERSION */
    int32_t num_envs;
    uint64_t seed;                  /* Philox key */
    /* sim (legged_robot_config.py:173-190) */
    float sim_dt;                   /* 0.005 */
    int32_t decimation;             /* 4 */
    float gravity_z;                /* -9.81 */
    int32_t solver_iterations;      /* PGS sweeps per substep */
    float contact_offset;           /* 0.01 */
    float max_depenetration_velocity; /* 1.0 */
    float ground_friction;          /* 1.0 (terrain.static_friction) */
    int32_t terrain_type;           /* 0 = plane, 1 = height field (the reference's 'heightfield' and 'trimesh' terrains) */
    /* control (go2_locomotion_config.py:53-60) */
    float kp, kd, action_scale, hip_scale_reduction, clip_actions;
    float default_dof_pos[QA_NUM_DOF];
    /* env */
    float env_spacing;              /* 3.0 */
    int32_t max_episode_length;     /* ceil(20 / 0.02) = 1000 */
    int32_t resampling_steps;       /* int(6 / 0.02) = 300 */
    int32_t push_interval;          /* ceil(8 / 0.02) = 400 */
    int32_t push_robots;
    float max_push_vel_xy;
    int32_t reset_mode;             /* 0 = default pose (legged_robot.py:581-596,614-634), 1 = mocap frames */
    float init_pos[3];              /* 0 0 0.42 */
    int32_t add_noise;
    /* noise_scale_vec entries (legged_robot.py:721-740): already multiplied by level and obs scale */
    float noise_roll_pitch, noise_ang_vel, noise_dof_pos, noise_dof_vel, noise_lin_vel;
    float clip_obs;                 /* 100 */
    /* obs scales (go2_locomotion_config.py:115-125) */
    float s_lin_vel, s_ang_vel, s_dof_pos, s_dof_vel, s_key_pos, s_foot_contact, s_lin_vel_dist, s_ang_vel_dist;
    /* rewards: scale*dt for each term, enum qa_reward order; 0 disables (legged_robot.py:927-932) */
    float reward_scale_dt[QA_NUM_REWARDS];
    int32_t only_positive_rewards;
    float tracking_sigma, soft_dof_pos_limit, soft_dof_vel_limit, soft_torque_limit, jump_goal;
    /* commands (go2_locomotion_config.py:165-181) */
    float lin_vel_x[QA_NUM_GAITS][2], lin_vel_y[QA_NUM_GAITS][2], ang_vel_yaw[QA_NUM_GAITS][2];
    float jump_height[2], locomotion_height[2];
    float lin_vel_x_clip, lin_vel_y_clip, ang_vel_yaw_clip;
    float latent_temperature;       /* 0.25 (legged_robot.py:536) */
    /* domain randomisation (go2_locomotion_config.py:74-100) */
    int32_t randomize_friction, randomize_base_mass, randomize_base_com, randomize_motor, use_easi;
    float friction_range[2], added_mass_range[2], added_com_range[2], motor_strength_range[2];
    float easi_mean[6], easi_var[6];
    /* height-field terrain (terrain_type 1; legged_robot_config.py:19-44, terrain.py:10-45) */
    int32_t hf_rows, hf_cols;       /* samples along x and y (tot_rows, tot_cols) */
    float hf_hscale, hf_vscale;     /* 0.1 m, 0.005 m */
    float hf_border;                /* border_size [m]: world x = row * hscale - border */
    float reset_xy_jitter;          /* custom_origins: default-pose resets add U(-j, j) to x and y (legged_robot.py:622-625) */
    /* mocap clips (reset_mode 1) */
    int32_t num_mocap_frames;       /* rows of QA_T_MOCAP_FRAMES */
    int32_t export_body_state;      /* != 0: step / simulate also refresh QA_T_RIGID_BODY_STATE (N,19,13) */
    int32_t env_id_offset;          /* GLOBAL id of local env 0: every random draw is keyed by (seed; env_id_offset + e, step, stream), and
                                       the spawn grid is laid out over global ids, so rank r of a data-parallel run that owns envs
                                       [r N/W, (r+1) N/W) of an N-env job reproduces exactly those envs of the one-process run */
    int32_t num_envs_global;        /* envs of the whole job (0 = num_envs): size of the spawn grid */
    int32_t contact_slots;          /* non-foot contacts per leg and substep.  2 (default; 0 means 2): every body group of the leg -- hip link /
                                       base share, thigh, calf -- has its own candidate (its lowest point) and up to two of the three
                                       make contact at once (all three inside the contact offset: the one with the largest gap waits),
                                       so _reward_collision and check_termination see the bodies independently; 1 = only the lowest
                                       non-foot point of the leg (the round-1 model) */
    int32_t hf_ceiling;             /* terrain_type 1: != 0 = QA_T_CEILING_SAMPLES holds overhangs to collide with */
    int32_t articulated_obstacles;  /* terrain_type 1: != 0 = the see-saw is a 1-DoF revolute plank with joint damping and the bar / tyre are
                                       1-DoF prismatic bodies under the reference's position drive, coupled to the contact rows
                                       (QA_T_OBST_DESC / QA_T_OBST_STATE) -- instead of the static shapes the height map draws for them */
    int32_t self_collision;         /* != 0: the lower legs (knee-to-foot capsules) of neighbouring legs collide -- left/right and front/rear pairs,
                                       one frictionless row per pair at the closest points; the reference's assets enable self-collision
                                       (`self_collisions = 0`, bbc/legged_gym/envs/go2/go2_locomotion_config.py:72,
                                       tsc/legged_gym/envs/go2/go2_agility_config.py:43).  The force is reported on the calf bodies. */
} qa_config;

typedef struct qa_sim qa_sim;

/* bytes the caller must allocate for the arena (256-byte aligned base). */
int64_t qa_arena_bytes(const qa_config *cfg);

/* Create a handle over a caller-owned DEVICE arena.  Replaces gym.create_sim / create_env /
 * create_actor / prepare_sim / acquire_*_tensor (legged_robot.py:351-364,995-1107,747-754).
 * Fills per-env parameters (friction buckets, added mass/CoM, motor strength, env origins)
 * from the Philox key and puts every env in the reset state (reset_buf = 1). */
int qa_create(const qa_config *cfg, void *arena, int64_t arena_bytes, void *stream, qa_sim **out);
int qa_destroy(qa_sim *sim);

/* Where tensor `which` lives inside the arena.  Replaces gymtorch.wrap_tensor. */
int qa_tensor_info(const qa_config *cfg, int which, int64_t *byte_offset, int64_t shape[3],
                   int32_t *ndim, int32_t *dtype);

/* The fused LeggedRobot.step(): action-history roll, optional delay (delay_steps in {0,1},
 * legged_robot.py:87-93), clip, `decimation` x (PD torque -> forward dynamics -> contact ->
 * integrate), post_physics_step (termination, 14 rewards, reset of done envs, observations).
 * `actions` is a device pointer to (N,12) fp32.  `global_step` is the env's step counter
 * before this call (common_step_counter); it keys the RNG and triggers pushes.
 * Launch shape (r5, no ABI change): one workgroup per 16 envs; on plane terrain, while the launch has at most one
 * workgroup per compute unit (N <= 16 x CUs = 4096 on an MI355X), every workgroup carries two HELPER wavefronts beside the
 * envs' own one (substep side chains, non-foot contact rows, the observation rows' history shift, the closing scalar stores);
 * same arithmetic, same outputs to rounding.  QA_ENV_HELPERS=0 / 1 in the environment forces them off / on. */
int qa_env_step(qa_sim *sim, const float *actions, int32_t delay_steps, int64_t global_step, void *stream);

/* qa_env_step for recorded launches: the step counter is read from DEVICE memory (one int64) and incremented by the
 * last workgroup of the step kernel to finish (arrival counter QA_T_STEP_TICKET), so a hipGraph holding N consecutive
 * steps replays without host-side arguments (the reference keeps `common_step_counter` on the host, legged_robot.py:134). */
int qa_env_step_dev(qa_sim *sim, const float *actions, int32_t delay_steps, int64_t *step_counter_dev, void *stream);

/* Training-mode exports (ABI 13).  Every step the reference's env refreshes tensors that nothing between two steps of a TRAINING run reads
 * (they serve seam 1, play / logging and the task-level tree): 920 B of the 4,680 B an env-step writes.  mask bit 0: qa_env_step
 * / qa_env_step_dev stop writing QA_T_CONTACT_FORCES, _RIGID_BODY_POS, _TORQUES, _TORQUES_ORG, _ACTIONS, _BASE_LIN_VEL, _BASE_ANG_VEL,
 * _PROJECTED_GRAVITY, _RPY, _FEET_FORCE, _CONTACT_FILT, _SCAN_HEIGHT (they keep their last values) and maintain only the two newest slots of
 * QA_T_ACTION_HISTORY (what a delay <= 1 can reach; a larger delay is refused while the mode is on).  mask bit 1 (only with bit 0: mask 3):
 * QA_T_OBS_DISC / _OBS_DISC_TERM are not written either (the AMP runner is their only reader).  Everything the learner consumes --
 * observations, rewards, resets, time-outs, commands, episode statistics, the simulator state -- is bit-identical with and without the mode.
 * mask 0 (default) = the reference's behaviour.  The CPU twin accepts the call and keeps exporting. */
int qa_set_lean_exports(qa_sim *sim, int32_t mask);

/* reset_idx(all) followed by nothing else (legged_robot.py:67-69).  The reference's reset()
 * then takes one zero-action step; the host mirror does that through qa_env_step. */
int qa_reset_all(qa_sim *sim, int64_t global_step, void *stream);

/* Seam-1 granular entry: one physics substep with caller-provided joint torques (N,12),
 * i.e. set_dof_actuation_force_tensor + simulate + fetch_results + refresh_*_tensor
 * (legged_robot.py:103-106,129-131).  Updates ROOT_STATES, DOF_STATE, CONTACT_FORCES,
 * RIGID_BODY_POS (and RIGID_BODY_STATE with cfg.export_body_state) only. */
int qa_simulate(qa_sim *sim, const float *torques, void *stream);

/* ---- task-level (TSC) env: the simulator half of tsc/legged_gym/envs/base/legged_robot.py ----------------------------------
 * The task-level env's step (:113-153) is the same physics as the behaviour-level one, but its post_physics_step is different
 * code (goals, 8 rewards, three observation rows: qa_tsc_goal_step / qa_tsc_observations below), so the fused kernel is also
 * offered without its post-physics phase:
 *
 * qa_env_physics_step = :118-139: action-history roll, delay (the task-level config always delays by one step, :124-126),
 * clip, decimation x (_compute_torques :762-795 -> gym.simulate), and the refresh_* of :231-234.  Updates ROOT_STATES,
 * DOF_STATE, CONTACT_FORCES, RIGID_BODY_POS, RIGID_BODY_STATE (cfg.export_body_state), TORQUES, TORQUES_ORG, ACTIONS,
 * ACTION_HISTORY, FOOT_IMPULSE; LAST_ACTIONS / LAST_DOF_VEL / LAST_TORQUES_ORG / LAST_ROOT_VEL take the values the step STARTS
 * from, which is what :275-278 left there at the end of the previous step.  Nothing else of the arena is touched.  The
 * obstacle course is the height-field terrain of the handle (the reference collides with separate obstacle actors). */
int qa_env_physics_step(qa_sim *sim, const float *actions, int32_t delay_steps, void *stream);

/* qa_tsc_reset = the simulator part of the task-level reset_idx (:348-410) for the envs with reset_flags[e] != 0 (device,
 * uint8): _reset_dofs (:796-807: joints at their default angles, at rest) and _reset_root_states (:840-884): root at
 * base_init_state with xy = start_xy[e] (the env's first goal, or the goal of a random obstacle with obstacle.randomize_start)
 * + rand_x_range U(-1,0), rand_y_range U(-1,1); orientation quat_from_euler_xyz(0, rand_pitch_range U(-1,1), start_yaw[e] +
 * rand_yaw_range U(-1,1)); velocities zero; last_actions / last_dof_vel / last_torques_org, the action history and the contact
 * warm start of those envs are cleared, RESET is set; last_root_vel is cleared for ALL envs (:389).  The uniforms come from the
 * handle's Philox generator keyed by (seed; env, global_step).  The goal / episode bookkeeping of the reset belongs to the
 * caller (TaskLevelBookkeeping.reset_idx). */
int qa_tsc_reset(qa_sim *sim, const uint8_t *reset_flags, const float *start_xy, const float *start_yaw, float rand_yaw_range,
                 float rand_x_range, float rand_y_range, float rand_pitch_range, int64_t global_step, void *stream);
/* the same with the step key read from device memory at execution time (a recorded rollout replays this launch: a by-value key
 * would repeat the same draws every replay) */
int qa_tsc_reset_dev(qa_sim *sim, const uint8_t *reset_flags, const float *start_xy, const float *start_yaw, float rand_yaw_range,
                     float rand_x_range, float rand_y_range, float rand_pitch_range, const int64_t *global_step_dev, void *stream);

/* The torch glue of the task-level env step between its kernels, as three launches (ABI 13; it was ~40 eager launches per env step inside the recorded
 * rollout).  All three key their draws like the engine: Philox (seed; env_id_offset + env, step, stream) with step = the device step counter.
 *
 * qa_tsc_push  =  `common_step_counter += 1` + `_push_robots` (tsc/legged_gym/envs/base/legged_robot.py:905-915): s = *step_dev + 1 is written back by the
 *   launch's last workgroup (`ticket`: one int32, zero before the first call, zero again after every call); on the steps with s % push_interval == 0
 *   (and push_interval > 0) root_states[e, 7:9] = (2 U - 1) max_push_vel_xy, U from stream 22.
 * qa_tsc_start_pose  =  the first lines of reset_idx (:352-366) for EVERY env (the reset kernel reads them where flagged): with randomize_start the
 *   envs with flags != 0 draw a new start obstacle (uniform over num_obstacles, stream 23) into cur_obst_idx; start_goal = cur_obst_idx * goals_per_obstacle
 *   (0 without randomize_start), start_xy = env_goals[e, start_goal, 0:2], start_yaw = obst_angs[e, cur_obst_idx] (frame_yaw0 without randomize_start).
 * qa_tsc_reset_where  =  reset_idx's bookkeeping (:367-376, 396-404) + `_reset_dofs`' obstacle part (:812-823), masked, no index list: for flags != 0
 *   cur_goal_idx = start_goal, reach_goal_timer = 0, episode_sums[:, e] = 0, episode_length = 0; for every env cur_goals / next_goals are re-gathered
 *   (indices clamped to the goal slots); with obst_state != NULL the flagged envs' see-saw goes to +-seesaw_rest (minus when cur_obst_idx >
 *   seesaw_order[e], seesaw_order != NULL) and every env's obstacle velocities are zeroed when *any_reset != 0. */
int qa_tsc_push(float *root_states, int64_t num_envs, int64_t *step_dev, int32_t *ticket, int32_t push_interval, float max_push_vel_xy, uint64_t seed,
                int32_t env_id_offset, void *stream);
int qa_tsc_start_pose(const uint8_t *flags, int64_t *cur_obst_idx, const float *env_goals, const float *obst_angs, int64_t num_envs, int32_t num_goal_slots,
                      int32_t num_obstacles, int32_t goals_per_obstacle, int32_t randomize_start, float frame_yaw0, uint64_t seed, const int64_t *step_dev,
                      int32_t env_id_offset, float *start_xy, float *start_yaw, int64_t *start_goal, void *stream);
int qa_tsc_reset_where(const uint8_t *flags, const uint8_t *any_reset, const int64_t *start_goal, int64_t *cur_goal_idx, float *reach_goal_timer,
                       float *episode_sums, int32_t num_terms, int64_t *episode_length, const float *env_goals, int32_t num_goal_slots, float *cur_goals,
                       float *next_goals, float *obst_state, float seesaw_rest, const int64_t *cur_obst_idx, const int64_t *seesaw_order, int64_t num_envs,
                       void *stream);

/* Reset bookkeeping of one task-level step in a single launch (tsc/legged_gym/envs/base/legged_robot.py:382-384, 396-404):
 *   any_reset[0]      = 1 if any reset_flags[e] != 0 (the condition of the reference's extra gym.simulate: feed it to qa_simulate_if)
 *   episode_means[k]  = sum over the flagged envs of episode_sums[k][e] / count / max_episode_length_s   for k < num_terms,
 *                       left untouched when no env is flagged (`extras["episode"]` keeps the last values)
 * episode_sums is (num_terms, num_envs) row-major.  One workgroup, fixed summation order (lane t adds envs t, t + 1024, ...; then a
 * binary tree): bit-reproducible, and bit-identical to the C twin. */
int qa_tsc_reset_stats(const uint8_t *reset_flags, const float *episode_sums, int64_t num_envs, int32_t num_terms, float max_episode_length_s,
                       float *episode_means, uint8_t *any_reset, void *stream);

/* qa_simulate only if *cond_dev != 0 (one device byte): the reference's reset_idx runs one more gym.simulate -- for EVERY env,
 * with the actuation forces set last -- whenever at least one env resets (:382-384).  torques == NULL applies QA_T_TORQUES. */
int qa_simulate_if(qa_sim *sim, const float *torques, const uint8_t *cond_dev, void *stream);

/* Upload the labelled mocap clips for reset_mode 1 (LeggedRobot.reset_idx with mocap_state_init,
 * bbc/legged_gym/envs/base/legged_robot.py:205-214, 598-612, 660-680; MotionLoader.get_full_frame_batch,
 * bbc/rsl_rl/datasets/motion_loader.py:461-474).  `frames` is a HOST pointer to (num_frames, QA_MOCAP_FRAME) fp32: the clips'
 * frames one after the other, each row root_pos3, root_quat4 (xyzw, normalised, w >= 0), joint_pos12, lin_vel3, ang_vel3
 * (both in the root frame), joint_vel12 -- i.e. columns [0:19] and [31:49] of the reference's 61-float frame after
 * MotionLoader.reorder.  `clips` is a HOST pointer to (num_clips <= QA_MAX_MOCAP_CLIPS, QA_MOCAP_CLIP) float64 rows, sorted
 * by gait; the clips of gait g are rows first_clip[g] .. first_clip[g+1]-1.  A reset of an env with gait g then does what the
 * reference does: clip ~ MotionWeight inside the gait (first clip whose cumulative probability exceeds u0), t = max(1e-7,
 * range u1) (traj_time_sample_batch, :333-342), p n = t / length n in float64, frames floor / ceil(p n), blend = p n -
 * floor, linear interpolation of everything but the root quaternion, which goes through the reference's quaternion_slerp
 * (bbc/rsl_rl/utils/utils.py:126-159, including its 1/angle weights); root velocities rotated into the world frame. */
int qa_set_mocap(qa_sim *sim, const float *frames, int32_t num_frames, const double *clips, int32_t num_clips,
                 const int32_t first_clip[QA_NUM_GAITS + 1], void *stream);

/* Verification entry (twin of the oracle's qo_debug_post_physics): LeggedRobot.post_physics_step
 * (bbc/legged_gym/envs/base/legged_robot.py:124-166) ALONE, on whatever state is in the arena -- the same device code the
 * fused step runs after its last substep, fed from ROOT_STATES / DOF_STATE / CONTACT_FORCES / RIGID_BODY_POS / TORQUES(_ORG) /
 * ACTIONS / ACTION_HISTORY instead of from registers.  tests/test_hip_parity.py replays the reference's own
 * post_physics_step fixtures (tests/golden/env_post_physics.npz) through it on the GPU. */
int qa_debug_post_physics(qa_sim *sim, int64_t global_step, void *stream);

/* Fused GAE (rollout_storage.py:97-111): reverse scan over T, then advantage
 * normalisation over all T*N samples (unbiased std, +1e-8).  All pointers are device
 * pointers; rewards/values/returns/advantages are (T,N) fp32, dones (T,N) uint8,
 * last_values (N).  `scratch` must hold at least 4096 bytes. */
int qa_gae(const float *rewards, const float *values, const uint8_t *dones, const float *last_values,
           float *returns, float *advantages, int32_t T, int32_t N, float gamma, float lam,
           int32_t normalize, void *scratch, void *stream);

/* ---- learner kernels ------------------------------------------------------------------------
 * Fused PPO minibatch objective and its gradient (SSInfoGAIL.update_actor_critic,
 * bbc/rsl_rl/algorithms/gail.py:333-345, 363-403; Normal log_prob/entropy as torch.distributions):
 *   logp   = sum_j -(a-mu)^2/(2 std^2) - log std - log sqrt(2 pi)          entropy = sum_j 1/2 + log sqrt(2 pi) + log std
 *   ratio  = exp(logp - old_logp);   surrogate = mean max(-A ratio, -A clamp(ratio, 1-clip, 1+clip))
 *   value  = mean max((v-R)^2, (tv + clamp(v-tv, -clip, clip) - R)^2)       (clipped_value != 0; else mean (R-v)^2)
 *   bound  = mean sum_j clamp(mu+1, max=0)^2 + clamp(mu-1, min=0)^2
 *   kl     = mean sum_j log(std/old_sigma + 1e-5) + (old_sigma^2 + (old_mu-mu)^2)/(2 std^2) - 1/2   (no gradient)
 *   loss   = c_surr surrogate + c_value value + c_bound bound - c_entropy entropy
 * Inputs (device, fp32): mu, actions, old_mu, old_sigma (B,12) row-major and 16-byte aligned; std (12);
 * value, old_logp, advantages, returns, target_values (B).  Outputs: dmu (B,12) = dloss/dmu, dstd (12),
 * dvalue (B), out[8] = {loss, surrogate, value, bound, entropy, kl, 0, 0}.  `scratch`: device memory of at least
 * qa_ppo_loss_scratch_bytes(B) bytes (per-wavefront partial sums; reduced in a fixed order, no atomics). */
int64_t qa_ppo_loss_scratch_bytes(int64_t B);
int qa_ppo_loss(const float *mu, const float *std, const float *value, const float *actions, const float *old_logp,
                const float *old_mu, const float *old_sigma, const float *advantages, const float *returns,
                const float *target_values, int64_t B, int32_t num_actions, float clip, float c_surr, float c_value,
                float c_bound, float c_entropy, int32_t clipped_value, float *dmu, float *dstd, float *dvalue, float *out,
                void *scratch, int64_t scratch_bytes, void *stream);

/* The HYBRID-action objective of the task-level learner and its gradient in one pass (tsc/rsl_rl/algorithms/ppo.py:222-262 with the
 * distributions of tsc/rsl_rl/modules/actor_critic.py:253-284): a categorical gait head and a Gaussian parameter head share the advantage,
 *   p = softmax(logits);  logp_d = log clamp(p[a_d], eps, 1 - eps)  (torch's Categorical(probs=p));  logp_c = sum_j log N(a_j; mean_j, std_j)
 *   surrogate = mean max(-A r_d, -A clip(r_d)) + mean max(-A r_c, -A clip(r_c)),   r_x = exp(logp_x - old_logp_x)
 *   entropy   = H(p) + (1 / num_c) sum_j H(N(., std_j))     (the Gaussian entropy is AVERAGED over its dimensions, :245-246)
 *   loss      = surrogate + c_value * value_loss(clipped as in qa_ppo_loss) - c_entropy * mean entropy
 *   kl        = mean sum_j [log(std_j / old_sigma_j + 1e-5) + (old_sigma_j^2 + (old_mu_j - mean_j)^2) / (2 std_j^2) - 1/2]   (Gaussian head only)
 * actions (B, 1 + num_c) = [gait index as a float | parameter vector].  Outputs: d loss / d logits (B, num_d), d mean (B, num_c),
 * d std (num_c), d value (B), out[8] = {loss, surrogate, value loss, entropy, kl, surrogate_d, surrogate_c, 0} (batch means).
 * num_d = 3 and num_c = 18 are built (Go2AgilityCfg: 3 gaits x 6 parameters); other widths return QA_E_ARG and the caller keeps its
 * eager expression.  `scratch` holds at least qa_hybrid_ppo_loss_scratch_bytes(B) bytes; fixed-order reductions, no atomics. */
int64_t qa_hybrid_ppo_loss_scratch_bytes(int64_t B);
int qa_hybrid_ppo_loss(const float *logits, const float *mean, const float *std, const float *value, const float *actions, const float *old_logp_d,
                       const float *old_logp_c, const float *old_mu, const float *old_sigma, const float *advantages, const float *returns,
                       const float *target_values, int64_t B, int32_t num_d, int32_t num_c, float clip, float c_value, float c_entropy,
                       int32_t clipped_value, float *dlogits, float *dmean, float *dstd, float *dvalue, float *out, void *scratch,
                       int64_t scratch_bytes, void *stream);

/* ELU backward fused with the bias-gradient column sum: the part of the backward of a Linear+ELU layer
 * (the `_mlp` blocks of bbc/rsl_rl/modules/actor_critic.py:92-139, estimator.py:12-33) that is not a GEMM:
 *   grad_in[r][c] = grad_out[r][c] * (out[r][c] > 0 ? 1 : out[r][c] + alpha),   grad_bias[c] = sum_r grad_in[r][c]
 * `out` is the layer's ELU OUTPUT (rows, cols) row-major; grad_in may not alias grad_out.  `scratch` holds at least
 * qa_elu_backward_bias_scratch_bytes(rows, cols) bytes.
 * grad_bias == NULL (ABI 14): the column sums are left IN PARTS -- on return `scratch` holds ceil(rows / 64) rows of `cols` floats, row b =
 * the column sums of grad_in's rows [64 b, 64 b + 64) -- for a consumer that adds them in row order itself (qa_clip_adam_step_reduce,
 * qa_grad_reduce); the finishing launch is not made. */
int64_t qa_elu_backward_bias_scratch_bytes(int64_t rows, int32_t cols);
int qa_elu_backward_bias(const float *grad_out, const float *out, float *grad_in, float *grad_bias, int64_t rows, int32_t cols,
                         float alpha, void *scratch, int64_t scratch_bytes, void *stream);

/* Weight and bias gradient of a NARROW linear layer -- the heads of the reference's networks: critic 128 -> 1, actor 128 -> 12
 * (bbc/rsl_rl/modules/actor_critic.py:127-139), the task-level gait / parameter heads 128 -> 3 / 18:
 *   grad_weight[o][k] = sum_r grad_out[r][o] * x[r][k],   grad_bias[o] = sum_r grad_out[r][o]        (out_features <= 32)
 * As a GEMM this is an (out x in) output with a rows-long reduction: one or a dozen output rows give the library nothing to
 * parallelise over (60 us for 1 x 128 over 24,576 rows, 22 us for 12 x 128) -- it is a weighted column sum of x, i.e. one streaming
 * pass.  Row slabs -> partial sums -> fixed-order finish (deterministic, no atomics).  grad_out (rows, out) and x (rows, in)
 * row-major; `scratch` holds at least qa_narrow_wgrad_scratch_bytes(rows, out, in) bytes. */
#define QA_NARROW_MAX_OUT 32
int64_t qa_narrow_wgrad_scratch_bytes(int64_t rows, int32_t out_features, int32_t in_features);
int qa_narrow_wgrad(const float *grad_out, const float *x, int64_t rows, int32_t out_features, int32_t in_features, float *grad_weight,
                    float *grad_bias, void *scratch, int64_t scratch_bytes, void *stream);

/* Dense layers of the learner's networks under training -- hand-written fp32-MFMA GEMMs (csrc/qa_gemm.hip; ABI 10).
 * They replace, for the Linear(+ELU/ReLU) blocks of bbc/rsl_rl/modules/actor_critic.py:92-139,171-225 and estimator.py:12-36 as
 * SSInfoGAIL.update_actor_critic (bbc/rsl_rl/algorithms/gail.py:328-413) and the task-level PPO.update
 * (tsc/rsl_rl/algorithms/ppo.py:160-282) run them, what PyTorch issues as addmm + elu (forward) and elu_backward + mm + mm^T + sum(0)
 * (backward).  All matrices fp32 row-major with explicit leading dimensions (so a column slice of a wider row -- e.g. the
 * proprioceptive block of the 671-wide observation -- is an operand without a copy); weight = nn.Linear.weight (out, in).
 * act: 0 none, 1 ELU(alpha), 2 ReLU.  fp32 in, fp32 accumulate: the reference's dtype.
 *   qa_linear_forward          y[r][o]  = act( sum_k x[r][k] weight[o][k] + bias[o] )                    (bias may be NULL)
 *   qa_linear_backward_input   gin[r][k] = ( sum_o gout[r][o] weight[o][k] ) * act'(y_prev[r][k])        act' from the activation OUTPUT
 *                              y_prev of the layer that produced x (ELU: y > 0 ? 1 : y + alpha; ReLU: y > 0; act_prev 0: no factor,
 *                              y_prev may be NULL) -- the elementwise half of the previous layer's backward, fused into this epilogue
 *   qa_linear_backward_weight  gw[o][k] = sum_r gout[r][o] x[r][k],  gb[o] = sum_r gout[r][o]            split over row slabs, partial
 *                              sums added in a fixed order (bit-reproducible, no atomics); gw is (out, in) contiguous;
 *                              `scratch` 16-byte aligned, at least qa_linear_backward_weight_scratch_bytes(rows, in, out) bytes. */
int qa_linear_forward(const float *x, int64_t ldx, const float *weight, int64_t ldw, const float *bias, float *y, int64_t ldy, int64_t rows,
                      int32_t in_features, int32_t out_features, int32_t act, float alpha, void *stream);
int qa_linear_backward_input(const float *grad_out, int64_t ldg, const float *weight, int64_t ldw, const float *y_prev, int64_t ldyp,
                             float *grad_in, int64_t ldgi, int64_t rows, int32_t in_features, int32_t out_features, int32_t act_prev, float alpha,
                             void *stream);
/* qa_linear_forward with the reduction (input-feature) dimension split over workgroups and the partial products added in a fixed order
 * before bias and activation: for layers with few outputs and very long rows (the depth encoder's Linear(62,400, 128),
 * tsc/rsl_rl/modules/depth_backbone.py:69, has 2 x rows/64 output tiles -- without the split 32-64 workgroups on 256 CUs).
 * out_features % 4 == 0, ldy % 4 == 0; scratch >= qa_linear_forward_split_scratch_bytes(...), 16-byte aligned.  (ABI 12) */
int64_t qa_linear_forward_split_scratch_bytes(int64_t rows, int32_t in_features, int32_t out_features);
int qa_linear_forward_split(const float *x, int64_t ldx, const float *weight, int64_t ldw, const float *bias, float *y, int64_t ldy, int64_t rows,
                            int32_t in_features, int32_t out_features, int32_t act, float alpha, void *scratch, int64_t scratch_bytes, void *stream);
int64_t qa_linear_backward_weight_scratch_bytes(int64_t rows, int32_t in_features, int32_t out_features);
/* ABI 17: qa_linear_backward_weight with grad_weight == NULL and grad_bias == NULL leaves the gradient IN PARTS in `scratch` (no reduction
 * launch): layout[0] weight parts, each in_features * out_features floats, layout[1] floats apart, from scratch + 0; layout[2] bias parts, each
 * out_features floats, layout[3] floats apart, from scratch + layout[4] floats -- what qa_clip_adam_step_reduce / qa_grad_reduce add in order. */
int qa_linear_backward_weight_layout(int64_t rows, int32_t in_features, int32_t out_features, int64_t layout[5]);
/* ABI 17: several weight (+ bias) gradient products in as few launches as their operands' alignments allow (one per combination of 16-byte /
 * 4-byte readable x and grad_out: at most four), each product what qa_linear_backward_weight computes -- the dozen small products of a chain
 * training step (csrc/qa_policy.hip, ABI 17) side by side instead of one after the other.  Per product: grad_weight and grad_bias both NULL leaves
 * the parts in `scratch` as qa_linear_backward_weight_batch_layout describes (same meaning as qa_linear_backward_weight_layout; the batch plans its
 * own split, so sizes and layout come from the _batch_ functions); otherwise every finished product of the batch is added up by ONE qa_grad_reduce
 * launch at the end.  Fixed work assignment and summation order (bit-reproducible).  When in_features is not a multiple of 4 but ldx leaves
 * room for it (ldx >= in_features rounded up to 4, x 16-byte aligned, ldx % 4 == 0) the rows are READ up to that multiple -- the columns behind
 * in_features must be readable memory of the same rows (any finite or non-finite values: they reach no output). */
typedef struct qa_wgrad_desc {
    const float *grad_out; int64_t ldg;        /* (rows, out_features), row stride ldg */
    const float *x; int64_t ldx;               /* (rows, in_features), row stride ldx */
    float *grad_weight, *grad_bias;            /* (out_features, in_features) contiguous, (out_features); both NULL: in parts */
    int64_t rows;
    int32_t in_features, out_features;
    void *scratch; int64_t scratch_bytes;      /* 16-byte aligned, >= qa_linear_backward_weight_batch_scratch_bytes(rows, in, out) */
} qa_wgrad_desc;
int64_t qa_linear_backward_weight_batch_scratch_bytes(int64_t rows, int32_t in_features, int32_t out_features);
int qa_linear_backward_weight_batch_layout(int64_t rows, int32_t in_features, int32_t out_features, int64_t layout[5]);
int qa_linear_backward_weight_batch(const qa_wgrad_desc *descs, int32_t count, void *stream);
int qa_linear_backward_weight(const float *grad_out, int64_t ldg, const float *x, int64_t ldx, float *grad_weight, float *grad_bias, int64_t rows,
                              int32_t in_features, int32_t out_features, void *scratch, int64_t scratch_bytes, void *stream);

/* out[i] = slabs[0][i] + slabs[1][i] + ... in that order (slab z at slabs + z * slab_stride): the fixed-order sum of split partial products
 * (the reduction half of qa_linear_backward_weight, exposed for partial products computed elsewhere).  Replaces torch's `sum(0)` over
 * the slab dimension in bbc/rsl_rl/algorithms/gail.py:328-413's weight gradients as this build evaluates them. */
int qa_slab_sum(const float *slabs, int64_t slab_stride, int32_t num_slabs, int64_t n, float *out, void *stream);

/* Image stem of the vision student's depth encoder under training (csrc/qa_conv.hip, csrc/qa_gemm.hip; ABI 12).  They replace, for
 * `DepthOnlyFCBackbone58x87.image_compression[0:5]` (tsc/rsl_rl/modules/depth_backbone.py:63-75: Conv2d(1, 32, 5) -> MaxPool2d(2, 2) -> ELU
 * -> Conv2d(32, 64, 3) -> ELU) as learn_vision / update_depth_actor / BYOL run it (tsc/rsl_rl/runners/on_policy_runner.py:278-441,
 * algorithms/ppo.py:327-358, modules/byol.py:242-317), what PyTorch issues as MIOpen convolution / pooling / aten elu kernels, forward
 * and backward.  Activations are CHANNELS-LAST fp32 ([image][y][x][channel]); weights of the generic convolution are
 * [out channel][ky][kx][in channel] (nn.Conv2d.weight permuted (0, 2, 3, 1)); act: 0 none, 1 ELU(alpha), 2 ReLU.
 *   qa_depth_stem_forward      y[n][py][px][c] = ELU( max over the 2x2 window of (conv5x5(images[n], weight[c]) + bias[c]) ), 32 channels,
 *                              images [n][ih][iw], weight [32][25] (= nn.Conv2d(1, 32, 5).weight), pooled size ((ih - 4) / 2, (iw - 4) / 2)
 *                              rounded down (MaxPool2d's floor mode); argmax[n][py][px][c] = 2 dy + dx of the window's FIRST maximum in
 *                              row-major order (PyTorch's tie rule), one byte each.  6 <= ih, 6 <= iw <= 126.
 *   qa_depth_stem_backward     grad_wb[0:800] = d loss / d weight, grad_wb[800:832] = d loss / d bias from grad_pre = the gradient at the
 *                              pooled pre-activation ([n][ph][pw][32]: what qa_conv_nhwc_backward_input with act_prev = 1 returns);
 *                              fixed work assignment and summation order; scratch >= qa_depth_stem_backward_scratch_bytes(), 16-byte aligned.
 *   qa_conv_nhwc_forward       y[n][oy][ox][o] = act( sum_{ky,kx,c} x[n][oy+ky][ox+kx][c] weight[o][ky][kx][c] + bias[o] )   ("valid", stride 1)
 *                              as one fp32-MFMA GEMM that reads the windows in place.  cin a power of two >= 16, cout % 4 == 0, more than
 *                              256 output pixels per image, all pointers 16-byte aligned.
 *   qa_conv_nhwc_backward_input  grad_in = conv(grad_padded, weight_flipped) * act'(x_act): the same GEMM over the output gradient
 *                              padded with kh - 1 / kw - 1 zeros on every side ([n][ihp][iwp][cout]: qa_elu_backward_pad writes it) and
 *                              weight_flipped[c][ky][kx][o] = weight[o][kh-1-ky][kw-1-kx][c]; x_act = the activation OUTPUT that was
 *                              this convolution's input ([n][ihp-kh+1][iwp-kw+1][cin]; act_prev 0: no factor, may be NULL).
 *   qa_conv_nhwc_backward_weight  grad_weight[o][ky][kx][c] = sum over images and output pixels of grad_out * x window, grad_bias[o] = sum
 *                              of grad_out; split over pixel slabs, fixed-order sum; scratch >= ..._scratch_bytes(...), 16-byte aligned.
 *   qa_elu_backward_pad        grad_pre = grad_out * act'(y) ([n][oh][ow][c], c % 4 == 0), written a second time into grad_pre_padded
 *                              [n][oh + 2 pad][ow + 2 pad][c] whose border is zeroed. */
int qa_depth_stem_forward(const float *images, const float *weight, const float *bias, float *y, uint8_t *argmax, int64_t n_img, int32_t ih, int32_t iw,
                          float alpha, void *stream);
int64_t qa_depth_stem_backward_scratch_bytes(void);
int qa_depth_stem_backward(const float *images, const uint8_t *argmax, const float *grad_pre, float *grad_wb, int64_t n_img, int32_t ih, int32_t iw,
                           void *scratch, int64_t scratch_bytes, void *stream);
int qa_conv_nhwc_forward(const float *x, const float *weight, const float *bias, float *y, int64_t n_img, int32_t ih, int32_t iw, int32_t cin,
                         int32_t kh, int32_t kw, int32_t cout, int32_t act, float alpha, void *stream);
int qa_conv_nhwc_backward_input(const float *grad_padded, const float *weight_flipped, const float *x_act, float *grad_in, int64_t n_img, int32_t ihp,
                                int32_t iwp, int32_t cout, int32_t kh, int32_t kw, int32_t cin, int32_t act_prev, float alpha, void *stream);
int64_t qa_conv_nhwc_backward_weight_scratch_bytes(int64_t n_img, int32_t ih, int32_t iw, int32_t cin, int32_t kh, int32_t kw, int32_t cout);
int qa_conv_nhwc_backward_weight(const float *x, const float *grad_out, float *grad_weight, float *grad_bias, int64_t n_img, int32_t ih, int32_t iw,
                                 int32_t cin, int32_t kh, int32_t kw, int32_t cout, void *scratch, int64_t scratch_bytes, void *stream);
int qa_elu_backward_pad(const float *grad_out, const float *y, float *grad_pre, float *grad_pre_padded, int64_t n_img, int32_t oh, int32_t ow,
                        int32_t channels, int32_t pad, int32_t act, float alpha, void *stream);

/* Running-moment normaliser of the discriminator inputs (bbc/rsl_rl/utils/utils.py:62-103).
 * qa_normalizer_update folds num_batches (1..4) row-major (rows[i], dim) fp32 device batches, in order, into the
 * device-resident double moments (mean[dim], var[dim], *count): each batch contributes its mean and biased variance
 * (RunningMeanStd.update_from_moments).  `batches` / `rows` are HOST arrays of device pointers / row counts; dim <= 128.
 * qa_normalizer_apply writes y = clamp((x - (float)mean) / sqrt((float)(var + epsilon)), -clip, clip). */
int qa_normalizer_update(const float *const *batches, const int64_t *rows, int32_t num_batches, int32_t dim,
                         double *mean, double *var, double *count, void *stream);
int qa_normalizer_apply(const float *x, float *y, int64_t rows, int32_t dim, const double *mean, const double *var,
                        float epsilon, float clip, void *stream);

/* Gradient clipping + Adam over a set of fp32 tensors in three launches (torch.nn.utils.clip_grad_norm_ followed by
 * torch.optim.Adam.step with amsgrad=False, maximize=False, L2 weight decay; gail.py:363-365, 403-405):
 *   coef = max_norm > 0 ? min(1, max_norm / (||g||_2 + 1e-6)) : 1;   g = coef grad + wd p;   t = steps[0][0] + 1
 *   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr / (1-b1^t) * m / (sqrt(v) / sqrt(1-b2^t) + eps)
 * params / grads / exp_avg / exp_avg_sq / steps are DEVICE arrays of num_tensors device pointers (steps: one float
 * counter per tensor, all set to t); the work list is num_chunks (tensor, start, length <= 2048) triples in device
 * memory; weight_decay is per tensor, lr one device float.  scratch >= 4 + num_chunks floats; scratch[3] = ||g||_2 (0 when
 * max_norm <= 0: the norm is then not computed and the step is two launches -- ONE when scratch has a fifth spare float,
 * scratch_floats >= 5 + num_chunks, zeroed before the first call: scratch[4 + num_chunks] is then the arrival counter of the
 * update kernel's workgroups, the last of which writes the step counters; the kernel leaves it at zero). */
int qa_clip_adam_step(float *const *params, const float *const *grads, float *const *exp_avg, float *const *exp_avg_sq,
                      float *const *steps, int32_t num_tensors, const int32_t *chunk_tensor, const int32_t *chunk_start,
                      const int32_t *chunk_len, int32_t num_chunks, const float *weight_decay, const float *lr, float beta1,
                      float beta2, float eps, float max_norm, float *scratch, int64_t scratch_floats, void *stream);
/* The same step with `grads_host` a HOST array of the num_tensors (<= QA_ADAM_MAX_INLINE) device gradient pointers: they travel
 * in the kernel arguments, so a step whose gradient tensors are re-allocated every time (autograd) needs no pointer-table copy. */
#define QA_ADAM_MAX_INLINE 64
int qa_clip_adam_step_hostgrads(float *const *params, const float *const *grads_host, float *const *exp_avg, float *const *exp_avg_sq,
                                float *const *steps, int32_t num_tensors, const int32_t *chunk_tensor, const int32_t *chunk_start,
                                const int32_t *chunk_len, int32_t num_chunks, const float *weight_decay, const float *lr, float beta1,
                                float beta2, float eps, float max_norm, float *scratch, int64_t scratch_floats, void *stream);
/* r5 (ABI 14): the same step for gradients that are still IN PARTS.  The weight gradient of a Linear layer leaves its split-K product as S
 * slabs (gail.py:328-413's backward through `torch.mm`; here fused.weight_grad's batched product) and its bias gradient leaves
 * qa_elu_backward_bias as one row of column sums per 64-row block: tensor t with red_parts_host[t] > 0 is
 *     grad[t][i] = sum_z red_src_host[t][z * red_stride_host[t] + i],  z = 0 .. parts - 1 in order (bit-reproducible)
 * and the sums-of-squares pass of the step adds the parts, WRITES the result to grads_host[t] (so whatever reads `.grad` after the step sees
 * the finished gradient) and squares it: the ~19 fixed-order finish launches of a PPO minibatch step (qa_colsum_finish, qa_slab_reduce) are
 * not made.  red_parts_host[t] == 0: the gradient is already final.  All three red_* arrays are HOST arrays of num_tensors entries
 * (<= QA_ADAM_MAX_INLINE); max_norm must be > 0 (the parts are added by the clipping pass).  Chunks of a tensor whose parts exceed 16 must
 * be at most 32 elements long (the eight thread rows of a workgroup share the parts): the caller's chunk table gives small tensors -- the
 * biases -- 32-element chunks. */
int qa_clip_adam_step_reduce(float *const *params, const float *const *grads_host, float *const *exp_avg, float *const *exp_avg_sq,
                             float *const *steps, int32_t num_tensors, const int32_t *chunk_tensor, const int32_t *chunk_start,
                             const int32_t *chunk_len, int32_t num_chunks, const float *weight_decay, const float *lr, float beta1,
                             float beta2, float eps, float max_norm, float *scratch, int64_t scratch_floats,
                             const float *const *red_src_host, const int64_t *red_stride_host, const int32_t *red_parts_host, void *stream);
/* ABI 18: TWO clipping optimisers and the KL rule between them in the same three launches.  A PPO minibatch step ends with
 * clip + Adam on the estimator, the KL-adaptive learning rate, clip + Adam on the actor-critic (bbc/rsl_rl/algorithms/gail.py:359-362, 367-379, 405-408;
 * tsc/rsl_rl/algorithms/ppo.py:243-262): seven launches of this library, ~35 us of a 370 us chain step.  The tables are those of
 * qa_clip_adam_step_reduce over BOTH optimisers' tensors, the first optimiser's first (tensors [0, split_tensor), chunks [0, split_chunk)); each half
 * has its own clipping norm (max_norm / pair->max_norm2, both > 0), step counter and learning rate (lr / pair->lr2, device floats); when pair->kl is
 * not NULL the finalize launch applies qa_kl_lr_rule(kl, desired_kl, kl_factor, lr_min, lr_max) to *lr2 before the update reads it.  The red_*
 * arrays may be NULL (every gradient final).  scratch: >= num_chunks + 9 floats (second head behind the single-launch step's counter). */
typedef struct qa_adam_pair {
    int32_t split_tensor, split_chunk;
    float *lr2; float max_norm2;
    const float *kl; float desired_kl, kl_factor, lr_min, lr_max;
} qa_adam_pair;
int qa_clip_adam_pair_step(float *const *params, const float *const *grads_host, float *const *exp_avg, float *const *exp_avg_sq,
                           float *const *steps, int32_t num_tensors, const int32_t *chunk_tensor, const int32_t *chunk_start,
                           const int32_t *chunk_len, int32_t num_chunks, const float *weight_decay, const float *lr, float beta1,
                           float beta2, float eps, float max_norm, float *scratch, int64_t scratch_floats,
                           const float *const *red_src_host, const int64_t *red_stride_host, const int32_t *red_parts_host, const qa_adam_pair *pair, void *stream);
/* The parts of up to QA_ADAM_MAX_INLINE gradients added in ONE launch without an optimiser step (the data-parallel step packs finished
 * gradients into its all-reduce bucket; tests): dst_host[t][i] = sum_z src_host[t][z stride_host[t] + i], i < numel_host[t]. */
int qa_grad_reduce(float *const *dst_host, const float *const *src_host, const int64_t *stride_host, const int32_t *parts_host,
                   const int32_t *numel_host, int32_t num_tensors, void *stream);

/* ABI 18: the optimiser half of a DISCRIMINATOR step in one launch.  The reference steps three Adam optimisers one after the other
 * (bbc/rsl_rl/algorithms/gail.py:518-520), each over the trunk's parameters plus one head's (gail.py:107-132): the trunk is stepped THREE times per
 * minibatch from the same gradient, each time with that optimiser's own moments and step counter, the later ones seeing the weights the earlier
 * ones left (their weight decay reads them).  An element's updates depend on nothing but that element, so one pass applies a tensor's states in
 * order; the gradient it starts from is put together in the same pass:
 *     grad[i] = sum_z src1[z stride1 + i]  (parts1 > 0; else grad[i] as it is)        the head losses' product, still in parts
 *             + alpha2 * sum_z src2[z stride2 + i]  (parts2 > 0; summed into tmp first)   the gradient penalty's product (gail.py:487-501)
 *             + reg * param[i]                                                       the weight regularisers 2 c W (gail.py:503-511)
 *     for s < num_states:  g = weight_decay_s * param[i] + grad[i];  Adam moments, bias corrections from step_s + 1, param[i] -= lr_s ...
 * grad is WRITTEN (what reads `.grad` after the step sees the finished gradient).  Sums in part order (bit-reproducible), fused multiply-adds
 * for the three additions.  Tensors with more than 16 parts in either source are walked in 32-element chunks (bias gradients), the others in
 * 512-element chunks.  All pointers are device pointers (lr included: one float each); `ticket` is one zeroed device word the launch leaves at
 * zero (the last workgroup to arrive writes the incremented step counters).  At most QA_ADAM_STACK_MAX_TENSORS tensors: they travel in the
 * kernel arguments. */
#define QA_ADAM_STACK_MAX_TENSORS 16
#define QA_ADAM_STACK_MAX_STATES 3
typedef struct qa_adam_stack_state { float *exp_avg, *exp_avg_sq, *step; const float *lr; float weight_decay; int32_t pad_; } qa_adam_stack_state;
typedef struct qa_adam_stack_tensor {
    float *param, *grad, *tmp;                 /* tmp: numel floats, needed when parts2 > 0 */
    const float *src1, *src2;
    int64_t stride1, stride2;
    int32_t parts1, parts2;
    float alpha2, reg;
    int32_t numel, num_states;
    qa_adam_stack_state state[QA_ADAM_STACK_MAX_STATES];
} qa_adam_stack_tensor;
int qa_adam_stack_step(const qa_adam_stack_tensor *tensors_host, int32_t count, float beta1, float beta2, float eps, uint32_t *ticket, void *stream);

/* ABI 18: acc[i] += *src_host[i] for i < count <= QA_ACC_MAX (src_host: a HOST array of device pointers to single floats): the six values a PPO
 * minibatch step logs (bbc/rsl_rl/algorithms/gail.py:275-283) come out of three kernels; this adds them onto the update's accumulator in one launch. */
#define QA_ACC_MAX 16
int qa_accumulate_scalars(float *acc, const float *const *src_host, int32_t count, void *stream);

/* Two small losses of the PPO step with their gradient in the same pass (a (rows, cols) contiguous, b (rows, cols) with row
 * stride b_stride, fp32 device pointers; grad_a (rows, cols); out[1]):
 *   QA_PAIR_ROW_L2: out = mean_r ||a_r - b_r||_2, grad_a = (a - b) / (||a_r - b_r|| rows), 0 where the norm is 0 -- the
 *                   privileged-latent regulariser (a = priv_encoder(latent), b = history latent, gail.py:346-350)
 *   QA_PAIR_MSE:    out = mean (a - b)^2, grad_a = 2 (a - b) / (rows cols) -- the estimator regression (gail.py:356-358) */
#define QA_PAIR_ROW_L2 0
#define QA_PAIR_MSE 1
int64_t qa_pair_loss_scratch_bytes(int64_t rows);
int qa_pair_loss(const float *a, const float *b, int64_t rows, int32_t cols, int64_t b_stride, int32_t mode, float *grad_a, float *out,
                 void *scratch, int64_t scratch_bytes, void *stream);
/* ABI 18: up to QA_PAIR_MAX_JOBS of those losses in ONE launch (the last workgroup to arrive adds each job's partial sums in qa_pair_loss's fixed order):
 * out[0] as qa_pair_loss gives it; grad_a the same, times *grad_scale when that device scalar is given (the regulariser's scheduled coefficient,
 * gail.py:353-357 -- the value in `out` stays unscaled, as the reference logs it).  scratch: >= qa_pair_losses_scratch_bytes(jobs, count), its FIRST
 * word zero before the first call (the launch leaves it at zero). */
#define QA_PAIR_MAX_JOBS 4
typedef struct qa_pair_job {
    const float *a, *b; int64_t rows; int32_t cols, mode; int64_t b_stride;
    const float *grad_scale;       /* NULL: 1 */
    float *grad_a, *out;
} qa_pair_job;
int64_t qa_pair_losses_scratch_bytes(const qa_pair_job *jobs, int32_t count);
int qa_pair_losses(const qa_pair_job *jobs, int32_t count, void *scratch, int64_t scratch_bytes, void *stream);

/* Minibatch gather (RolloutStorage.mini_batch_generator, bbc/rsl_rl/storage/rollout_storage.py:122-157): for t < num_tensors,
 * dst[t][r, 0:widths[t]] = src[t][idx[b rows + r], 0:widths[t]] with b = *idx_block (a DEVICE scalar; NULL = 0: idx is then
 * just `rows` long); src rows are src_strides[t] floats apart, dst rows are dense.  `src`, `src_strides`, `widths`, `dst` are HOST
 * arrays of device pointers / sizes; idx int64 on the device.  The block form lets a recorded step walk through a table of
 * pre-drawn sample indices (one block per step) with a device-side step counter. */
#define QA_GATHER_MAX 12
int qa_gather_rows(const int64_t *idx, const int64_t *idx_block, int64_t rows, int32_t num_tensors, const float *const *src,
                   const int64_t *src_strides, const int32_t *widths, float *const *dst, void *stream);

/* KL-adaptive learning rate of the PPO step (bbc/rsl_rl/algorithms/gail.py:367-379) on DEVICE scalars, so that a recorded
 * step never reads the KL on the host:  *lr = max(lr_min, *lr / factor) if *kl > 2 desired_kl;  min(lr_max, *lr * factor) if
 * 0 < *kl < desired_kl / 2;  unchanged otherwise.  The reference uses factor 1.5, lr_min 1e-5, lr_max 1e-2. */
int qa_kl_lr_rule(const float *kl, float desired_kl, float factor, float lr_min, float lr_max, float *lr, void *stream);

/* extras["episode"] of reset_idx (bbc/legged_gym/envs/base/legged_robot.py:229-240: per reward term, the mean over the envs that reset this
 * step of episode_sum / max_episode_length_s) from QA_T_EPISODE_STATS without a host sync and in ONE launch (ABI 13; it was ten eager
 * launches per env step inside the recorded rollout): bin = (step - 1) & 1 with step = *step_dev when step_dev != NULL (the DEVICE step
 * counter AFTER the step; recorded rollouts) else `step`; cnt = stats[bin][14]; for i < num_terms (<= 14):
 *   means[i] = cnt > 0 ? stats[bin][i] / cnt / max_episode_length_s : means[i]   (kept when nobody reset, as the reference keeps its dict)
 *   snapshot[i] = means[i]                                                         (the per-step values the runner averages over a rollout) */
int qa_episode_means(const float *episode_stats, const int64_t *step_dev, int64_t step, int32_t num_terms, float max_episode_length_s, float *means,
                     float *snapshot, void *stream);

/* Rollout bookkeeping around qa_env_step (SSInfoGAIL.act / process_env_step, bbc/rsl_rl/algorithms/gail.py:176-212;
 * RolloutStorage.add_transitions, rollout_storage.py:60-74; the runner's episode sums, on_policy_runner.py:187-206).
 * qa_rollout_act: a = mean + std * eps, log-prob of a under N(mean, std); writes `actions` (N,12) for the env and the
 *   step's storage rows st_actions / st_mu / st_sigma (N,12), st_logp / st_values (N).  `noise` (N,12) supplies eps;
 *   NULL draws it from the engine's Philox generator, stream 20, keyed by (seed; env, step) with step = *step_dev when
 *   step_dev != NULL (recorded launches) else `step`; env = env_id_offset + row (the engine's global env id, qa_config).
 * qa_rollout_post: st_rewards = reward_coef * rew + gamma * values * time_out; st_dones = reset > 0; if `cur` != NULL,
 *   the six running sums cur (6,N) [total, i, us, ss, t, length] advance by [reward_coef*rew, 0, 0, 0, rew, 1], are
 *   copied to fin_vals (6,N), and are cleared where done; fin_mask (N) = done. */
int qa_rollout_act(const float *mean, const float *std, const float *value, const float *noise, uint64_t seed, const int64_t *step_dev,
                   int64_t step, int32_t num_envs, int32_t env_id_offset, float *actions, float *st_actions, float *st_mu, float *st_sigma, float *st_logp,
                   float *st_values, void *stream);
/* ABI 18: qa_rollout_act, and in the same launch the step's observation rows into the storage (RolloutStorage.add_transitions' `observations[step].copy_`,
 * rollout_storage.py:60-74): st_obs[r][0 .. obs_width) = obs[r][0 .. obs_width), row strides st_obs_stride / obs_stride (floats). */
int qa_rollout_act_store(const float *mean, const float *std, const float *value, const float *noise, uint64_t seed, const int64_t *step_dev,
                         int64_t step, int32_t num_envs, int32_t env_id_offset, float *actions, float *st_actions, float *st_mu, float *st_sigma, float *st_logp,
                         float *st_values, const float *obs, int64_t obs_stride, int32_t obs_width, float *st_obs, int64_t st_obs_stride, void *stream);
/* qa_rollout_act_hybrid (ABI 13): the task-level teacher's sampling and bookkeeping of one env step (tsc/rsl_rl/modules/actor_critic.py:252-261
 * `act` / `get_actions_log_prob_d` / `_c`, tsc/rsl_rl/algorithms/ppo.py:101-125, RolloutStorage.add_transitions, and the runner's action history,
 * tsc/rsl_rl/runners/on_policy_runner.py:199-203) in one launch.  logits (N, nd), mean (N, nc_all), std (nc_all), value (N):
 *   p = softmax(logits);  a_d ~ Categorical(p) by inverse CDF on one Philox uniform (stream 21, keyed like qa_rollout_act);
 *   logp_d = log(clamp(p[a_d], eps, 1 - eps)), eps = 2^-23 (torch's Categorical(probs=...) through probs_to_logits);
 *   a_c = mean + std * N(0,1) (stream 20);  logp_c = sum_j log N(a_c_j; mean_j, std_j);
 *   actions (N, 1 + nc_all) = [a_d as float | a_c], also to st_actions; st_mu = mean, st_sigma = std, st_logp_d, st_logp_c, st_values = value;
 *   action_history (N, hist_len, 1 + nc_all), if not NULL, = action_history_in rolled by one slot (oldest first) with its newest slot set to
 *   `actions`; action_history_in NULL or == action_history rolls in place (slower: a thread per env), a second buffer is copied coalesced.
 * nd <= 16, nc_all <= 32. */
int qa_rollout_act_hybrid(const float *logits, const float *mean, const float *std, const float *value, uint64_t seed, const int64_t *step_dev, int64_t step,
                          int32_t num_envs, int32_t env_id_offset, int32_t nd, int32_t nc_all, float *actions, float *st_actions, float *st_mu, float *st_sigma,
                          float *st_logp_d, float *st_logp_c, float *st_values, const float *action_history_in, float *action_history, int32_t hist_len,
                          void *stream);
int qa_rollout_post(const float *rew, const int64_t *reset, const uint8_t *time_out, const float *values, float reward_coef, float gamma,
                    int32_t num_envs, float *st_rewards, uint8_t *st_dones, float *cur, float *fin_vals, uint8_t *fin_mask, void *stream);
/* The same with the discriminator's rewards (Discriminator.predict_disc_reward, bbc/rsl_rl/algorithms/discriminator.py:88-118,
 * MSELoss mapping): d, eps (N) and class logits (N, dim_c <= 8) are the three heads' outputs for this step's frame pair; the
 * labels are read from the step's observation rows obs (N rows, obs_stride floats apart): eps label = obs[num_obs - dim_c - 1],
 * class label = argmax obs[num_obs - dim_c :].
 *   p = max(softmax(logits), 1e-20);  r_i = max(0, 1 - (d - 1)^2 / 4) dt;  r_us = -|eps - label| dt;
 *   r_ss = (p[label] - logsumexp(p)) dt   (cross-entropy applied to the already-softmaxed p, as the reference)
 *   total = c_i r_i + c_us r_us + c_ss r_ss + c_t rew;   st_rewards = total + gamma values time_out
 * and the running sums advance by [total, r_i, r_us, r_ss, rew, 1]. */
int qa_rollout_post_amp(const float *rew, const int64_t *reset, const uint8_t *time_out, const float *values, const float *d, const float *eps,
                        const float *logits, int32_t dim_c, const float *obs, int64_t obs_stride, int32_t num_obs, float c_i, float c_us,
                        float c_ss, float c_t, float dt, float gamma, int32_t num_envs, float *st_rewards, uint8_t *st_dones, float *cur,
                        float *fin_vals, uint8_t *fin_mask, void *stream);

/* Head losses of the SS-InfoGAIL discriminator step and their gradient w.r.t. the head outputs
 * (bbc/rsl_rl/algorithms/gail.py:452-520, MSELoss variant).  Rows are [labelled expert b_lb | policy b_pi | unlabelled
 * expert b_ulb]; d, eps (rows), c (rows,5: the softmax the discriminator returns; entries below 1e-20 are clamped to 1e-20 here and
 * pass no gradient, i.e. discriminator.py:52's torch.clamp may be left to this kernel) are the three heads' outputs;
 * label_lb (b_lb) int64 gait labels; policy_eps (b_pi), policy_c (b_pi,5) the latents the policy samples were generated with.
 *   ss = mean_lb CE(log_softmax(c), label);  info = mean_ulb -sum c log(c + 1e-20);
 *   disc = 1/2 (mean_pi (d+1)^2 + mean_ulb (d-1)^2);  us = mean_pi |eps - policy_eps|
 *   loss = c_ss ss + (*info_coef_dev) info + c_disc disc + c_us us
 * Outputs: grad_d, grad_eps (rows), grad_c (rows,5) = d loss / d heads; out[16] = {loss, ss, info, disc, us, acc_lb
 * (argmax c == label), acc_pi (d < 0), acc_exp (d > 0 on ulb), acc_ulb (argmax c == argmax policy_c), mean_ulb c [5], 0, 0}.
 * The gradient penalty and the weight regularisers of the step are not part of this entry point. */
int64_t qa_disc_loss_scratch_bytes(int64_t rows);
int qa_disc_loss(const float *d, const float *eps, const float *c, const int64_t *label_lb, const float *policy_eps, const float *policy_c,
                 int32_t b_lb, int32_t b_pi, int32_t b_ulb, float c_ss, const float *info_coef_dev, float c_disc, float c_us,
                 float *grad_d, float *grad_eps, float *grad_c, float *out, void *scratch, int64_t scratch_bytes, void *stream);
/* ABI 18: the same with the class LOGITS in place of c: the softmax of Discriminator.forward (discriminator.py:66; max, exp, sum, divide) and its backward
 * happen inside -- grad_logits = d loss / d logits = p (g - <g, p>) with g the gradient w.r.t. the clamped probabilities as above; a discriminator chain step
 * made three launches of them (softmax, the product grad x output, the softmax backward).  Everything else, `out` included, as qa_disc_loss. */
int qa_disc_loss_logits(const float *d, const float *eps, const float *logits, const int64_t *label_lb, const float *policy_eps, const float *policy_c,
                        int32_t b_lb, int32_t b_pi, int32_t b_ulb, float c_ss, const float *info_coef_dev, float c_disc, float c_us,
                        float *grad_d, float *grad_eps, float *grad_logits, float *out, void *scratch, int64_t scratch_bytes, void *stream);

/* Sampling front of a discriminator step: what three `feed_forward_generator`s (bbc/rsl_rl/storage/replay_buffer.py:38-47,
 * bbc/rsl_rl/datasets/motion_loader.py feed_forward_generator_lb / _ulb) hand to update_ss_info_gail, drawn from index tables made once per
 * update, and qa_disc_prepare's arithmetic applied on the way: rows index[b][*block_dev * rows[b] + r] of src[b] (b = 0 labelled expert,
 * 1 policy replay ring, 2 unlabelled expert) land prepared in out ((rows[0] + rows[1] + rows[2]), dim), in that order; the policy rows'
 * latent targets eps_out[r] = eps_src[row], c_out[r][:] = c_src[row][:] (c_dim columns) and the labelled rows' classes
 * label_out[r] = label_src[*block_dev * rows[0] + r] alongside (eps_src / label_src may be NULL).  One launch; `block_dev` is a DEVICE
 * scalar so that a recorded step can be replayed for successive minibatches.  (ABI 12) */
typedef struct {
    const float *src[3];
    const int64_t *index[3];
    int64_t rows[3];
    const float *eps_src, *c_src;
    float *eps_out, *c_out;
    const int64_t *label_src;
    int64_t *label_out;
    const int64_t *block_dev;
} qa_disc_sample_io;
int qa_disc_sample_prepare(const qa_disc_sample_io *io, int32_t dim, int32_t c_dim, const float *task_mask, const float *frame_mult,
                           const float *task_weight_dev, const double *mean, const double *var, float epsilon, float clip, float *out, void *stream);

/* Tail of a discriminator step (bbc/rsl_rl/algorithms/gail.py:486-504, 520-533): the logged values that are sums of squares, assembled with
 * qa_disc_loss's head statistics into the 11 values update_ss_info_gail returns
 *   out = {ss, info_max, disc, us  (head_stats[1..4]),  sum(input_grad^2) / grad_rows  (gradient penalty),  sum(weights[last]^2)  (logit
 *          regulariser),  sum over all weights of sum(w^2)  (weight decay),  acc_lb, acc_pi, acc_exp, acc_ulb  (head_stats[5..8])}
 * and, if given, acc[0..10] += out, *step_counter += 1 (the recorded step's accumulator and device-side step counter) and the class prior's
 * EMA prior[k] = prior[k] (1 - prior_soft_coef) + prior_soft_coef head_stats[9 + k], k < prior_dim <= 5 (:463-464).  One launch, partial
 * sums added in a fixed order.  `weights` / `weight_counts` are HOST arrays of 1..7 device pointers / element counts.  `scratch`:
 * qa_disc_step_tail_scratch_bytes() bytes, 16-byte aligned, ZEROED once by the caller (it holds the arrival counter, which the kernel
 * leaves at zero).  (ABI 12) */
int64_t qa_disc_step_tail_scratch_bytes(void);
int qa_disc_step_tail(const float *head_stats, const float *input_grad, int64_t grad_rows, int32_t grad_cols, const float *const *weights,
                      const int64_t *weight_counts, int32_t num_weights, float *out, float *acc, int64_t *step_counter, float *prior, int32_t prior_dim,
                      float prior_soft_coef, void *scratch, int64_t scratch_bytes, void *stream);

/* Discriminator input preparation (bbc/rsl_rl/algorithms/discriminator.py:77-87, utils.py:97-103): 1..3 row-major
 * (rows[i], dim) fp32 device batches are written one under the other into out (sum rows, dim):
 *   y = ((x * (task_mask[c] ? *task_weight_dev : 1)) * frame_mult[c] - (float)mean[c]) / sqrt((float)(var[c] + epsilon)), clipped to +-clip
 * task_weight_dev == NULL skips the task weighting, mean == var == NULL the normalisation.  `batches` / `rows` are HOST arrays. */
int qa_disc_prepare(const float *const *batches, const int64_t *rows, int32_t num_batches, int32_t dim, const float *task_mask,
                    const float *frame_mult, const float *task_weight_dev, const double *mean, const double *var, float epsilon, float clip,
                    float *out, void *stream);

/* Policy inference: a chain of fully connected layers evaluated per 16-row tile with the activations in LDS (one launch
 * for Estimator.forward + ActorCritic.update_distribution + ActorCritic.evaluate of SSInfoGAIL.act,
 * bbc/rsl_rl/algorithms/gail.py:176-197; estimator.py:35-36; actor_critic.py:171-196,222-225 -- and the same modules
 * behind play.py / the exported policy).  The chain is a HOST array of ops run in order on every tile:
 *   QA_MLP_COPY : dst_buf[:, dst_col : dst_col+n] = src_buf[:, src_col : src_col+n]
 *   QA_MLP_LAYER: y = src_buf[:, src_col : src_col+k] @ W^T + b, W (n,k) row-major as nn.Linear stores it; act 1 applies
 *                 ELU(alpha 1), act 2 ReLU, act 3 tanh (ABI 13: the task-level scan encoder's last layer).  y goes to dst_buf[:, dst_col : dst_col+n], or with dst_buf = -1 to rows of the global
 *                 output `out_index`.
 * Buffer 0 holds the tile of the input rows x (x_cols <= QA_MLP_BUF0_COLS) and is read-only; buffers 1..3 are scratch
 * with QA_MLP_BUFn_COLS columns, zero at the start of the chain.  A layer's src_col must be a multiple of 4 and its
 * source and destination buffers must differ.  fp32 throughout (fp32 MFMA accumulation; the K order differs from a
 * library GEMM, so results agree with torch to rounding, not bit for bit).
 * qa_mlp_pack repacks the layers' weights into `packed` (w_off / b_off: float offsets chosen by the caller, w_off a
 * multiple of 4; qa_mlp_packed_floats gives the size the chosen offsets need); it must be re-run after the weights change.
 * weights[i] / biases[i] are DEVICE pointers for op i (ignored for copies; a NULL bias is zero).
 * ABI 17 -- the same launch as the two chain halves of a TRAINING step (bbc/rsl_rl/algorithms/gail.py:328-413 through
 * autograd: the forward of every network of a PPO minibatch step, then the input-gradient path):
 *   QA_MLP_F_SAVE        the op's result ALSO goes to outs[out_index][:, out_col : out_col+n] (layers into a scratch buffer, copies): the
 *                        activations the backward pass and the weight-gradient GEMMs read.  Every global write starts at column out_col.
 *   QA_MLP_F_TRANSPOSED  the matrix handed to qa_mlp_pack for this op is the FORWARD layer's own (k, n) row-major weight; the op
 *                        multiplies by its transpose: gx = g W (an input-gradient layer; give it a NULL bias).
 *   act 4 / 5 / 6        the product is SCALED by the derivative of ELU(alpha 1) / ReLU / tanh taken from the forward layer's saved
 *                        output y = outs[aux_index][:, aux_col : aux_col+n] (y > 0 ? 1 : y + 1;  y > 0;  1 - y^2) -- the elementwise half of
 *                        the layer below's backward, in this layer's epilogue.  `outs[aux_index]` is only read.
 *   QA_MLP_GRAD          elementwise: dst[:, dst_col : +n] = (src[:, src_col : +n] (+ dst with QA_MLP_F_ADD)) * act'(y) with act 0 (no
 *                        factor) or 4..6; QA_MLP_F_SAVE as above.  Where two gradients meet at an activation (the privileged encoder's output
 *                        feeds the actor AND the regulariser).
 *   QA_MLP_LOAD          dst[:, dst_col : +n] = outs[aux_index][:, aux_col : aux_col+n]: a second (third, ...) input of the chain read from global memory
 *                        into a scratch buffer (the gradients at several heads arrive as separate tensors); QA_MLP_F_SAVE writes the loaded columns
 *                        out again at (out_index, out_col), e.g. to assemble them into one matrix for a weight-gradient product.
 * Ops written before ABI 17 (new fields zero) mean what they meant. */
#define QA_MLP_COPY 0
#define QA_MLP_LAYER 1
#define QA_MLP_GRAD 2
#define QA_MLP_LOAD 3
#define QA_MLP_F_SAVE 1
#define QA_MLP_F_TRANSPOSED 2
#define QA_MLP_F_ADD 4
#define QA_MLP_MAX_OPS 24
#define QA_MLP_MAX_OUTPUTS 8      /* ABI 17 (4 before) */
#define QA_MLP_BUF0_COLS 800      /* the task-level policy's 800-wide observation row (ABI 13; 672 before) */
#define QA_MLP_BUF1_COLS 576
#define QA_MLP_BUF2_COLS 320
#define QA_MLP_BUF3_COLS 128
typedef struct qa_mlp_op {
    int32_t kind;
    int32_t src_buf, src_col;
    int32_t dst_buf, dst_col;
    int32_t k, n;
    int32_t act;
    int32_t out_index;
    int32_t flags;                         /* QA_MLP_F_* (ABI 17; `reserved`, zero, before) */
    int64_t w_off, b_off;
    int32_t out_col;                       /* ABI 17: first column of outs[out_index] the op writes */
    int32_t aux_index, aux_col;            /* ABI 17: act 4..6: the saved activation */
    int32_t pad_;
} qa_mlp_op;
int64_t qa_mlp_packed_floats(const qa_mlp_op *ops, int32_t num_ops);
int qa_mlp_pack(const qa_mlp_op *ops, int32_t num_ops, const float *const *weights, const float *const *biases, float *packed,
                int64_t packed_floats, void *stream);
int qa_mlp_forward(const float *x, int64_t x_stride, int32_t rows, int32_t x_cols, const qa_mlp_op *ops, int32_t num_ops,
                   const float *packed, float *const *outs, const int64_t *out_strides, int32_t num_outs, void *stream);
/* r5 (ABI 15): launches with few row tiles.  A launch of <= 128 tiles (2048 rows) leaves CUs idle while each busy one walks the whole chain, so
 * qa_mlp_forward gives every tile to up to min(4, 256 / tiles) workgroups, each running the ops of one STRAND: ops are grouped by flow dependence
 * through the scratch buffers (an op belongs with the last writer of every scratch column it reads; the input tile is shared), the groups dealt to
 * strands largest first onto the least loaded (cost = k * n per layer + a constant per op).  SSInfoGAIL.act has two groups (critic | estimator ->
 * privileged / history encoder -> actor).  Per-op arithmetic is unchanged, so outputs are bit-identical to the unsplit launch.  QA_MLP_STRANDS=1
 * in the environment switches the split off.  qa_mlp_strands is the (host-only) partition itself: strand_of[i] = strand of op i, return value =
 * number of strands used (1 = not split), or a negative QA_E_* code. */
int qa_mlp_strands(const qa_mlp_op *ops, int32_t num_ops, int32_t max_strands, int32_t *strand_of);
/* r5 (ABI 16): launches with a tile per CU or more (> 2048 rows).  There is nothing to split over CUs, and one workgroup walking the chain layer by
 * layer pays a cold weight stream, an epilogue and a workgroup barrier per layer (~30 % of a tile's cycles for SSInfoGAIL.act's 13 layers) while the
 * CU does nothing else.  When the chain has two strands (qa_mlp_strands with max_strands 2) and their buffers fit the CU's 160 KB of LDS, qa_mlp_forward
 * runs them SIDE BY SIDE in every workgroup: wavefronts 0..3 strand 0, wavefronts 4..7 strand 1, each group with its own copy of scratch buffers 1..3
 * (as wide as its ops touch them) behind the shared input tile, and its own barrier.  Per-tile arithmetic is unchanged => bit-identical outputs.
 * qa_mlp_groups is the (host-only) plan: strand_of[num_ops]; base / stride [2][4] = LDS offset and row stride (floats) of buffer b as group g sees it
 * (stride 0: the group does not touch the buffer); *lds_floats = the launch's LDS size.  Returns 1 when the two-group launch applies to this chain,
 * 0 when it does not (one strand, or no fit), negative QA_E_* on a malformed chain.  MEASURED SLOWER than the one-group kernel (87.7 v 85.5 us at 4096
 * rows): it ships off.  qa_mlp_set_groups(2) (or QA_MLP_GROUPS=2 in the environment) turns the two-group launch on, (1) off again; returns the
 * previous setting. */
int qa_mlp_groups(const qa_mlp_op *ops, int32_t num_ops, int32_t x_cols, int32_t *strand_of, int32_t *base, int32_t *stride, int32_t *lds_floats);
int qa_mlp_set_groups(int32_t groups);

/* ---------------------------------------------------------------------------------------------------------------
 * Task-level (TSC) env-side math of SURVEY 8a row a18: the two per-step pieces of tsc/legged_gym/envs/base/legged_robot.py
 * that need no obstacle physics.  The obstacle-course simulation itself is not part of this library.
 *
 * qa_tsc_set_commands  =  LeggedRobot.set_commands (tsc/legged_gym/envs/base/legged_robot.py:699-760): turn the task policy's
 * hybrid action (N, 1 + num_d * num_c) = [behaviour id, num_d x num_c parameters] into the behaviour policy's command block.
 * For envs with episode_length % interval == 0:  g = mocap_index[id];  p = clip(params of behaviour `id`, -1, 1);
 * latent_c = one_hot(g, dim_c);  latent_eps = p[num_c - 1];  u = (p + 1) / 2;  commands[0..2] = lo[g] + (hi[g] - lo[g]) u[0..2]
 * (per-gait ranges, `vel_ranges` (3, dim_c, 2): lin_vel_x, lin_vel_y, ang_vel_yaw);  commands[3] = jump range(u[3]) if g is the
 * last gait else 0;  commands[4] = locomotion-height range(u[4]) if g is not the last gait else 0.  Then, if `noise` is not
 * NULL, commands (ALL envs) *= noise (N,5) (domain_rand.randomize_action; the reference draws U(0.8,1.2) with torch's
 * generator, here the caller supplies the draw).  next_commands (N, 5 + 1 + dim_c) = [commands, latent_eps, latent_c].
 * num_c must be 6 and dim_c <= 8.  `mocap_index` (num_d) and the ranges are HOST arrays. */
int qa_tsc_set_commands(const float *actions, const int64_t *episode_length, int64_t num_envs, int32_t num_d, int32_t num_c, int32_t dim_c,
                        int32_t interval, const int32_t *mocap_index, const float *vel_ranges, const float *jump_range,
                        const float *height_range, const float *noise, float *commands, float *latent_eps, float *latent_c,
                        float *next_commands, void *stream);

/* qa_tsc_goal_step  =  the post-physics bookkeeping of the task-level env between `refresh_*` and `reset_idx`
 * (tsc/legged_gym/envs/base/legged_robot.py:226-262): episode_length += 1; body-frame velocities, projected gravity and
 * roll/pitch/yaw (:239-246, euler_from_quaternion :32-55); filtered foot contacts (:247-249); _update_goals (:204-224);
 * current obstacle type (:255-258); check_termination (:322-346); compute_reward (:412-430) with the 8 active terms of
 * legged_robot_config.py:307-332 in the reference's (alphabetical) summation order -- QA_TSC_REW_*, `reward_scales` already
 * multiplied by dt as _prepare_reward_function does (:1107-1113) -- clipped at zero before the termination term is added;
 * finally the goal gather of :272-273 for the (possibly advanced) goal index.  One thread per env.
 * All pointers are device pointers; `cur_goals` / `next_goals` are read first (they hold the previous step's gather) and
 * rewritten last.  Envs that the caller then resets must have their goal index zeroed and goals re-gathered by the reset. */
#define QA_TSC_REW_ACTION_HL_RATE 0
#define QA_TSC_REW_COLLISION 1
#define QA_TSC_REW_FEET_EDGE 2
#define QA_TSC_REW_LATENT_C_RATE 3
#define QA_TSC_REW_REACH_GOAL 4
#define QA_TSC_REW_TRACKING_GOAL_VEL 5
#define QA_TSC_REW_TRACKING_YAW 6
#define QA_TSC_REW_TERMINATION 7
#define QA_TSC_NUM_REWARDS 8
#define QA_TSC_MAX_BODY_IDS 24
typedef struct qa_tsc_goal_cfg {
    int64_t num_envs;
    int32_t num_bodies;                 /* rows of contact_forces / rigid_body_states per env, <= 32; body lists hold each body once */
    int32_t num_goal_slots;             /* env_goals.size(1), the repeated last goal included */
    int32_t last_goal_repeat, goals_per_obstacle, num_obstacles;
    int32_t history_len, history_width; /* action_hl_history (N, history_len, history_width); history_len >= 3 */
    int32_t mask_rows, mask_cols;       /* x_edge_mask */
    int32_t use_camera;                 /* cfg.depth.use_camera: also reset on reaching the last goal */
    int32_t num_termination_bodies, num_penalised_bodies;
    int32_t termination_bodies[QA_TSC_MAX_BODY_IDS], penalised_bodies[QA_TSC_MAX_BODY_IDS], feet_bodies[4];
    float reach_goal_delay_steps;       /* cfg.env.reach_goal_delay / dt */
    float next_goal_threshold, leave_goal_threshold;
    float max_episode_length;
    float target_lin_vel;               /* cfg.rewards.target_lin_vel; obstacle types 0 and 4 use 2.5 */
    float border_size, horizontal_scale;
    float reward_scales[QA_TSC_NUM_REWARDS];
} qa_tsc_goal_cfg;
typedef struct qa_tsc_goal_io {
    /* in */
    const float *root_states;           /* (N,13) pos, quat xyzw, linvel, angvel (world) */
    const float *contact_forces;        /* (N,num_bodies,3) */
    const float *rigid_body_states;     /* (N,num_bodies,13) */
    const float *env_goals;             /* (N,num_goal_slots,3) */
    const int64_t *obstacle_types;      /* (N,num_obstacles) */
    const float *action_hl_history;     /* (N,history_len,history_width) or NULL (the two rate terms are then 0) */
    const uint8_t *x_edge_mask;         /* (mask_rows,mask_cols) */
    /* in/out */
    int64_t *episode_length;            /* (N) */
    int64_t *cur_goal_idx;              /* (N) */
    float *reach_goal_timer;            /* (N) */
    uint8_t *last_contacts;             /* (N,4) */
    float *cur_goals, *next_goals;      /* (N,3) */
    float *episode_sums;                /* (QA_TSC_NUM_REWARDS,N) */
    /* out */
    float *base_lin_vel, *base_ang_vel, *projected_gravity, *rpy;   /* (N,3) each */
    uint8_t *contact_filt;              /* (N,4) */
    float *target_pos_rel, *next_target_pos_rel;                    /* (N,2) */
    float *target_yaw, *next_target_yaw;                            /* (N) */
    uint8_t *reached_goal;              /* (N) */
    int64_t *cur_obstacle_type;         /* (N) */
    uint8_t *reset_buf, *time_out_buf, *reach_goal_cutoff;          /* (N) */
    float *rew_buf;                     /* (N) */
} qa_tsc_goal_io;
int qa_tsc_goal_step(const qa_tsc_goal_cfg *cfg, const qa_tsc_goal_io *io, void *stream);

/* qa_tsc_observations  =  the task-level env's 132-point height scan and observation assembly
 * (tsc/legged_gym/envs/base/legged_robot.py: _get_heights :1708-1755 with quat_apply_yaw of legged_gym/utils/math.py:10-14,
 * compute_observations :432-515, compute_flat_key_pos :1929-1947), run after qa_tsc_goal_step on its outputs.  Layouts:
 *   proprio (57)   = [roll, pitch | base_ang_vel * ang_vel (3) | (dof_pos - default_dof_pos_all) * dof_pos (12) | dof_vel * dof_vel (12) |
 *                     last action (12) | contact_filt - 0.5 (4) | zeros (12)]
 *   obs_buf (800)  = [proprio 57 | delta_yaw, delta_next_yaw | one_hot(cur_obstacle_type, 6) | clip(z - 0.3 - heights, -1, 1) (132) |
 *                     root_h_obs, base_lin_vel * lin_vel (4) | mass params 4, friction 1, motor_strength[0]-1 (12), [1]-1 (12) |
 *                     obs_history BEFORE this step's push (10 x 57, oldest first)]
 *   obs_bbc (671)  = [proprio 57 | the 4 + 29 privileged values | the same history 570 | commands 5, latent_eps 1, latent_c 5]
 *   obs_disc (49)  = [roll, pitch, root_h, base_lin_vel * lin_vel_dist, base_ang_vel * ang_vel_dist, (dof_pos - default_dof_pos) *
 *                     dof_pos, dof_vel * dof_vel, key-body positions in the heading frame * key_pos (12), contact_filt * foot_contact]
 * root_h = z - heights[132 / 2 + 1].  Then the history push (:497-505): all 10 slots = proprio where episode_length <= 1, else shift
 * by one and append; obs_buf, obs_bbc and the history are clipped to +-clip_observations.  delta yaws are recomputed from
 * target_yaw - yaw (wrapped to [-pi, pi)) when `update_yaw` is set and carried over otherwise (:445-450).  The reference's
 * contact_buf roll (:507, read by nothing active) is not kept.  One wavefront per env; all pointers are device pointers. */
#define QA_TSC_NUM_SCAN 132
#define QA_TSC_NUM_PROPRIO 57
#define QA_TSC_HISTORY_LEN 10
#define QA_TSC_NUM_OBS 800
#define QA_TSC_NUM_OBS_BBC 671
#define QA_TSC_NUM_OBS_DISC 49
#define QA_TSC_NUM_OBSTACLE_CLASSES 6
typedef struct qa_tsc_obs_cfg {
    int64_t num_envs;
    int32_t num_bodies;                 /* rows of rigid_body_states per env */
    int32_t key_bodies[4];              /* the reference's key_body_ids (the feet) */
    int32_t map_rows, map_cols;         /* height_samples */
    int32_t update_yaw;                 /* global_counter % depth.update_interval == 0 */
    int32_t root_height_obs;            /* cfg.env.root_height_obs */
    int64_t action_stride;              /* floats between consecutive envs' rows of `last_action` (action_history_buf[:, -1]) */
    int64_t points_env_stride;          /* floats between consecutive envs' scan grids in `height_points`; 0 = one grid for all */
    int32_t point_stride, reserved;     /* floats between consecutive points (2 for xy, 3 for the reference's xyz tensor) */
    float border_size, horizontal_scale, vertical_scale;
    float lin_vel, ang_vel, dof_pos, dof_vel, lin_vel_dist, ang_vel_dist, key_pos, foot_contact;   /* obs_scales */
    float clip_observations;
    float default_dof_pos[12], default_dof_pos_all[12];
} qa_tsc_obs_cfg;
typedef struct qa_tsc_obs_io {
    /* in */
    const float *root_states;           /* (N,13) */
    const float *rpy, *base_lin_vel, *base_ang_vel;     /* (N,3), from qa_tsc_goal_step */
    const uint8_t *contact_filt;        /* (N,4) */
    const float *dof_pos, *dof_vel;     /* (N,12) */
    const float *last_action;           /* (N,12) with row stride action_stride */
    const float *rigid_body_states;     /* (N,num_bodies,13) */
    const float *mass_params;           /* (N,4) */
    const float *friction;              /* (N,1) */
    const float *motor_strength;        /* (2,N,12) */
    const int64_t *cur_obstacle_type;   /* (N) */
    const float *target_yaw, *next_target_yaw;          /* (N) */
    const int16_t *height_samples;      /* (map_rows,map_cols) */
    const float *height_points;         /* body-frame xy of the QA_TSC_NUM_SCAN scan points, strides in the cfg */
    const float *commands, *latent_eps, *latent_c;      /* (N,5), (N,1), (N,5) */
    const int64_t *episode_length;      /* (N) */
    /* in/out */
    float *delta_yaw, *delta_next_yaw;  /* (N) */
    float *obs_history;                 /* (N,10,57) */
    /* out */
    float *measured_heights;            /* (N,132) */
    float *obs_buf, *obs_bbc_buf, *obs_disc_buf;        /* (N,800), (N,671), (N,49) */
} qa_tsc_obs_io;
int qa_tsc_observations(const qa_tsc_obs_cfg *cfg, const qa_tsc_obs_io *io, void *stream);

/* qa_tsc_depth_update  =  the depth camera of the vision student: update_depth_buffer + process_depth_image
 * (tsc/legged_gym/envs/base/legged_robot.py:154-200) with the camera of attach_camera (:1203-1226) and the `depth` block of
 * legged_robot_config.py:63-84.  The reference rasterises the scene with Isaac Gym's camera sensor (106 x 60 depth image, 87 deg
 * horizontal FOV, camera at `position` on the trunk, pitched down by U(angle) degrees per env), then per env: crop [1:-1, 10:-9]
 * -> 58 x 87, clip to [near, far], normalise to (d - near) / (far - near) - 0.5, add noise, and push the image into a
 * `buffer_len`-deep ring (all slots = the image where episode_length <= 1).  This entry ray-casts the same image against what this
 * engine collides with: the course's height field and its ceiling field (QA_T_HEIGHT_SAMPLES / QA_T_CEILING_SAMPLES layouts).
 *   pixel (r, c) of the FULL image -> camera-frame ray (1, -((c + .5) / W * 2 - 1) tan(hfov / 2), -((r + .5) / H * 2 - 1) tan(hfov / 2) H / W);
 *   camera frame = trunk frame * R_y(pitch) at trunk position + R_trunk * position; depth = the ray parameter (distance along the
 *   optical axis, what a depth image stores);
 *   march: steps of dt = 0.5 hscale / max(|d_xy|, 0.5) up to `far`; the floor is hit where z - floor(x, y) turns negative, a ceiling
 *   where z - ceiling(x, y) changes sign between two samples that both have one (thin shell, seen from either side); the
 *   bracket is halved QA_TSC_DEPTH_BISECT times (a vertical face is a one-cell ramp in a height field: without this a wall reads up
 *   to one march step near) and the hit placed by linear interpolation inside it; no hit = far.
 *   noise (process_depth_image :166-168): image += depth_noise * 2 (u1 - .5) + (depth_noise u0) * 2 (u_px - .5), Philox stream 64
 *   keyed by (seed; env_id_offset + env, step): block 0 = (u0, u1), pixel p = component p & 3 of block 1 + (p >> 2).
 * Only the cropped pixels are cast.  One thread per pixel; all pointers are device pointers. */
#define QA_TSC_DEPTH_STREAM 64
#define QA_TSC_DEPTH_BISECT 4
typedef struct qa_tsc_depth_cfg {
    int64_t num_envs;
    int64_t step;                        /* RNG key: the env's global step counter */
    uint64_t seed;
    int32_t env_id_offset;
    int32_t width, height;               /* depth.original = (106, 60) */
    int32_t crop_top, crop_bottom, crop_left, crop_right;   /* 1, 1, 10, 9 -> (height - 2) x (width - 19) = 58 x 87 */
    int32_t buffer_len;                  /* depth.buffer_len = 2 */
    int32_t map_rows, map_cols;
    int32_t coarse_log2;                 /* S: the coarse maps of the io block summarise blocks of 2^S x 2^S cells (3 = 40 cm blocks of 5 cm cells) */
    float horizontal_fov_deg;            /* 87 */
    float position[3];                   /* camera position in the trunk frame */
    float near_clip, far_clip, depth_noise;
    float border_size, horizontal_scale, vertical_scale;
} qa_tsc_depth_cfg;
typedef struct qa_tsc_depth_io {
    const float *root_states;            /* (N,13) */
    const float *camera_pitch;           /* (N) radians, positive = down */
    const int16_t *height_samples;       /* (map_rows,map_cols) */
    const int16_t *ceiling_samples;      /* same grid, QA_NO_CEILING where none; NULL = no overhangs */
    const int64_t *episode_length;       /* (N) */
    float *depth_buffer;                 /* in/out (N, buffer_len, height - crop_top - crop_bottom, width - crop_left - crop_right) */
    /* optional acceleration structure (NULL = march every sample): block (I, J) = cells [I 2^S, (I + 1) 2^S) x [J 2^S, (J + 1) 2^S),
     * (((map_rows - 2) >> S) + 1, ((map_cols - 2) >> S) + 1) int16 each: the highest height sample / the lowest ceiling sample among
     * the (2^S + 1)^2 samples those cells touch.  It only lets the march skip samples that cannot report a hit: the image does not
     * depend on it (the CPU twin ignores it). */
    const int16_t *coarse_floor_max, *coarse_ceiling_min;
    const int64_t *step_dev;             /* optional: the noise key's step read from device memory at execution time instead of cfg.step */
} qa_tsc_depth_io;
int qa_tsc_depth_update(const qa_tsc_depth_cfg *cfg, const qa_tsc_depth_io *io, void *stream);

const char *qa_last_error(void);
int qa_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* QA_SIM_H */
