"""Go2 on the agility course (tsc/legged_gym/envs/go2/go2_agility_config.py:4-57): robot-specific overrides of the task-level
config tree -- names and values are the reference's."""
from quadrupedal_agility_amd.tsc.legged_gym.envs.base.legged_robot_config import LeggedRobotCfg, LeggedRobotCfgPPO

_LEGS = ("FL", "FR", "RL", "RR")


class Go2AgilityCfg(LeggedRobotCfg):
    class init_state(LeggedRobotCfg.init_state):
        pos = [0.0, 0.0, 0.42]
        default_joint_angles = {f"{l}_{j}_joint": a for l in _LEGS for j, a in (("hip", 0.0), ("thigh", 0.9), ("calf", -1.8))}

    class control(LeggedRobotCfg.control):
        control_type = "P"
        stiffness = {"joint": 40.0}
        damping = {"joint": 1}
        action_scale = 0.25
        action_bias_scale = 0.1
        hip_scale_reduction = 0.5
        decimation = 4

    class asset(LeggedRobotCfg.asset):
        file = "{LEGGED_GYM_ROOT_DIR}/resources/robots/go2/urdf/go2.urdf"
        foot_name = "foot"
        penalize_contacts_on = ["base", "Head_upper", "Head_lower"] + [f"{l}_{p}" for l in ("FL", "FR", "RL", "RR") for p in ("hip", "thigh", "calf")]
        terminate_after_contacts_on = ["base", "Head_upper", "Head_lower", "hip", "thigh"]
        self_collisions = 0

    class rewards(LeggedRobotCfg.rewards):
        soft_dof_pos_limit = 0.9
        base_height_target = 0.25


class Go2AgilityCfgPPO(LeggedRobotCfgPPO):
    class algorithm(LeggedRobotCfgPPO.algorithm):
        entropy_coef = 0.01

    class runner(LeggedRobotCfgPPO.runner):
        run_name = ""
        experiment_name = "agility"
        bbc_path = "weights/bbc/model.pt"
