"""Configuration tree of the task-level (TSC) env and its PPO learner.  Class names, field names and default values are the
reference's (tsc/legged_gym/envs/base/legged_robot_config.py:8-470): they are the contract `train.py`, the env, the course
generator and the runner read.  Fields nothing on this build's path reads (viewer, trimesh parkour terrain, camera intrinsics of
the depth student) are kept only where the reference's own code touches them at construction time."""
from quadrupedal_agility_amd.legged_gym.envs.base.base_config import BaseConfig

_OBSTACLES = ("bar_jump", "frame", "poles", "seesaw", "tire_jump", "tunnel")


class LeggedRobotCfg(BaseConfig):
    class play:
        load_student_config = False
        mask_priv_obs = False

    class env:
        num_envs = 6144
        n_scan, n_priv, n_delta_yaw, n_obst_type, n_priv_latent = 132, 4, 2, 6, 29
        n_auxiliary = n_delta_yaw + n_obst_type
        n_proprio = 57 + n_auxiliary
        history_len = 10
        mocap_category = ["trot", "canter", "jump"]
        mocap_category_all = ["walk", "pace", "trot", "canter", "jump"]
        num_actions_d = len(mocap_category)
        num_actions_c = 5 + 1
        num_actions_bbc = 12
        num_command = num_actions_c + len(mocap_category)
        num_observations = n_proprio + n_scan + n_priv_latent + n_priv + history_len * (n_proprio - n_auxiliary)
        num_observations_bbc = n_proprio - n_auxiliary + n_priv_latent + n_priv + num_actions_c + len(mocap_category_all)
        num_privileged_obs = None
        num_obs_disc = 49
        disc_obs_len = 2
        send_timeouts = True
        episode_length_s = 40
        history_encoding = True
        include_foot_contacts = True
        env_spacing = 3.0
        randomize_start_pos = False
        randomize_start_vel = True
        randomize_start_yaw, rand_yaw_range = True, 0.2
        randomize_start_x, rand_x_range = True, 0.2
        randomize_start_y, rand_y_range = True, 0.1
        randomize_start_pitch, rand_pitch_range = False, 1.6
        contact_buf_len = 100
        next_goal_threshold = 0.4
        reach_goal_delay = 0.02
        num_future_goal_obs = 2
        leave_goal_threshold = 4.0
        root_height_obs = True

    class depth:
        use_camera = False
        camera_num_envs = 256
        position = [0.305, 0.0175, 0.098]
        angle = [-5, 5]
        update_interval = 1
        original, resized = (106, 60), (87, 58)
        horizontal_fov = 87
        buffer_len = 2
        near_clip, far_clip, depth_noise = 0.3, 4, 0.05
        scale, invert = 1, True

    class normalization:
        class obs_scales:
            lin_vel, ang_vel, dof_pos, dof_vel = 0.5, 0.25, 1.0, 0.05
            key_pos = foot_contact = lin_vel_dist = ang_vel_dist = 0.0
            height_measurements = 5.0
        clip_observations = 100.0
        clip_actions = 100.0

    class noise:
        add_noise = False
        noise_level = 1.0
        quantize_height = True

        class noise_scales:
            rotation, dof_pos, dof_vel, lin_vel, ang_vel, gravity, height_measurements = 0.0, 0.01, 0.05, 0.05, 0.05, 0.02, 0.02

    class terrain:
        mesh_type = "obstacle"
        horizontal_scale, vertical_scale, border_size = 0.05, 0.005, 5
        curriculum = True
        static_friction = dynamic_friction = 1.0
        restitution = 0.0

    class obstacle:
        obstacle_dict = dict(zip(_OBSTACLES, (0.2, 0.15, 0.2, 0.15, 0.2, 0.1)))
        obstacle_proportions = list(obstacle_dict.values())
        num_links = dict(zip(_OBSTACLES, (2, 1, 1, 2, 2, 1)))
        num_obstacle_links = list(num_links.values())
        num_joints = dict(zip(_OBSTACLES, (1, 0, 0, 1, 1, 0)))
        num_obstacle_joints = list(num_joints.values())
        bar_jump_range, tire_jump_range = [0.05, 0.20], [0.40, 0.55]
        curriculum, curr_step, curr_threshold = False, 0.01, 0.8
        bar_jump_init_range, tire_jump_init_range = [0.05, 0.10], [0.40, 0.45]
        bar_jump_max_range, tire_jump_max_range = [0.05, 0.3], [0.40, 0.65]
        horizontal_scale, vertical_scale, border_size = 0.05, 0.005, 5
        border_height, randomize_border, border_height_range = 4.28, True, [0.0, 4.5]
        env_length, env_width, env_boarder = 7, 10, 1.5
        robot_org = [4.5, 0.5]
        num_goals, last_goal_repeat = 4, 2
        measure_heights = True
        measured_points_x = [0.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0, 1.1]
        measured_points_y = [-0.5, -0.4, -0.3, -0.2, -0.1, 0.0, 0.1, 0.2, 0.3, 0.4, 0.5]
        randomize_start = False
        num_obst_per_env = 6
        random_x = {k: [-0.25, 0.25] for k in _OBSTACLES}
        random_y, random_yaw = [-0.15, 0.15], [-5, 5]
        frame_pos = [[[5.5, 1.0], [5.5, 5.0]], [[5.5, 5.0], [5.5, 9.0]], [[3.5, 9.0], [3.5, 5.0]],
                     [[3.5, 5.0], [3.5, 1.0]], [[1.5, 1.0], [1.5, 5.0]], [[1.5, 5.0], [1.5, 9.0]]]
        frame_ang = [90, 90, -90, -90, 90, 90]

    class commands:
        curriculum, max_curriculum = False, 1.0
        num_commands = 5
        resampling_time = 0.02
        heading_command = True

        class ranges:          # rows: walk, pace, trot, canter, jump
            lin_vel_x = [[0.0, 0.6], [0.5, 1.5], [0.5, 1.5], [0.8, 2.5], [0.8, 2.0]]
            lin_vel_y = [[-0.15, 0.15], [-0.3, 0.3], [-0.3, 0.3], [-0.5, 0.5], [-0.3, 0.3]]
            ang_vel_yaw = [[-1.0, 1.0], [-1.57, 1.57], [-1.57, 1.57], [-0.5, 0.5], [-0.5, 0.5]]
            jump_height = [0.45, 0.58]
            locomotion_height = [0.25, 0.34]

    class init_state:
        pos = [0.0, 0.0, 1.0]
        rot = [0.0, 0.0, 0.0, 1.0]
        lin_vel = [0.0, 0.0, 0.0]
        ang_vel = [0.0, 0.0, 0.0]
        default_joint_angles = {"joint_a": 0.0, "joint_b": 0.0}

    class control:
        control_type = "P"
        stiffness = {"joint_a": 10.0, "joint_b": 15.0}
        damping = {"joint_a": 1.0, "joint_b": 1.5}
        action_scale = 0.5
        action_bbc_weight = 0.8
        hip_scale_reduction = 0.5
        decimation = 4

    class asset:
        file = ""
        foot_name = "None"
        penalize_contacts_on = []
        terminate_after_contacts_on = []
        collapse_fixed_joints = True
        fix_base_link = False
        self_collisions = 0

    class domain_rand:
        randomize_friction, friction_range = True, [0.6, 2.0]
        randomize_base_mass, added_mass_range = False, [0.0, 1.5]
        randomize_base_com, added_com_range = False, [-0.1, 0.1]
        push_robots, push_interval_s, max_push_vel_xy = False, 8, 0.5
        randomize_action, action_noise = True, [0.8, 1.2]
        randomize_motor, motor_strength_range = False, [0.8, 1.2]
        action_delay, action_delay_step, action_buf_len = True, 1, 8

    class rewards:
        class scales:
            termination = -50.0
            reach_goal = 5.0
            every_step = 0.0
            tracking_goal_vel = 0.4
            tracking_yaw = 2.0
            lin_vel_z = ang_vel_xy = orientation = dof_acc = 0.0
            collision = -20.0
            action_rate = 0.0
            action_hl_rate = -0.2
            latent_c_rate = -1.0
            delta_torques = torques = hip_pos = dof_error = feet_stumble = 0.0
            feet_edge = -1.0
            torque_limits = dof_pos_limits = dof_vel_limits = 0.0
        only_positive_rewards = True
        tracking_sigma = 0.25
        soft_dof_pos_limit, soft_dof_vel_limit, soft_torque_limit = 1.0, 1, 0.4
        base_height_target = 1.0
        max_contact_force = 40.0
        target_lin_vel = 0.4

    class viewer:
        ref_env = 0
        pos, lookat = [10, 0, 6], [11.0, 5, 3.0]

    class sim:
        dt = 0.005
        substeps = 1
        gravity = [0.0, 0.0, -9.81]
        up_axis = 1

        class physx:
            num_threads, solver_type = 10, 1
            num_position_iterations, num_velocity_iterations = 4, 0
            contact_offset, rest_offset = 0.01, 0.0
            bounce_threshold_velocity = 0.5
            max_depenetration_velocity = 1.0

        class qa:                       # this build's own solver knob (not in the reference)
            solver_iterations = 4


class LeggedRobotCfgPPO(BaseConfig):
    seed = 1
    runner_class_name = "OnPolicyRunner"

    class policy:
        init_noise_std = 1.0
        continue_from_last_std = True
        scan_encoder_dims = [128, 64, 32]
        actor_hidden_dims = [512, 256, 128]
        critic_hidden_dims = [512, 256, 128]
        priv_encoder_dims = [64]
        activation = "elu"
        tanh_encoder_output = False

    class algorithm:
        value_loss_coef = 1.0
        use_clipped_value_loss = True
        clip_param = 0.2
        entropy_coef = 0.01
        num_learning_epochs, num_mini_batches = 5, 4
        learning_rate = 5.0e-4
        schedule = "adaptive"
        gamma, lam = 0.99, 0.95
        desired_kl = 0.01
        max_grad_norm = 1.0
        dagger_update_freq = 20
        priv_reg_coef_schedual = [0, 0.1, 500, 1000]
        priv_reg_coef_schedual_resume = [0, 0.1, 0, 1]

    class depth_encoder:
        if_depth = LeggedRobotCfg.depth.use_camera
        depth_shape = LeggedRobotCfg.depth.resized
        buffer_len = LeggedRobotCfg.depth.buffer_len
        hidden_dims = 512
        learning_rate, learning_rate_byol, learning_rate_min = 1.0e-3, 3.0e-4, 1.0e-5
        num_steps_per_env = LeggedRobotCfg.depth.update_interval * 24

    class estimator:
        train_with_estimated_states = True
        learning_rate = 1.0e-4
        hidden_dims = [128, 64]
        priv_states_dim = LeggedRobotCfg.env.n_priv
        num_prop = LeggedRobotCfg.env.n_proprio - LeggedRobotCfg.env.n_auxiliary
        num_auxiliary = LeggedRobotCfg.env.n_auxiliary
        num_scan = LeggedRobotCfg.env.n_scan
        load_estimator_bbc = True

    class runner:
        policy_class_name = "ActorCritic"
        algorithm_class_name = "PPO"
        num_steps_per_env = 24
        max_iterations = 50000
        save_interval = 100
        experiment_name = "agility"
        run_name = ""
        resume = False
        load_run, checkpoint = -1, -1
        resume_path = None
        bbc_path = ""
        disc_loss_function = "MSELoss"
        reward_i_coef, reward_us_coef, reward_ss_coef, reward_t_coef = 0.05, 0.0, 0.0, 2.0
        disc_hidden_units = [512, 256]
