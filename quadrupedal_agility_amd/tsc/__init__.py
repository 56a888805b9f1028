"""TSC (task-level controller) tree of the reference.  Learner side: the hybrid categorical + Gaussian
policy `ActorCriticTSC`, the frozen low-level `ActorCriticBBC`, the hybrid `PPO` and its rollout storage (`tsc.rsl_rl`).
Env side: the command mapping, the goal / termination / reward bookkeeping, the 132-point height scan and the observation
assembly of the task-level `LeggedRobot` (`tsc.legged_gym.TaskLevelBookkeeping`, HIP kernels `qa_tsc_set_commands` /
`qa_tsc_goal_step` / `qa_tsc_observations`).  The obstacle-course simulation itself (obstacle contact, see-saw joints, depth
camera) is NOT built -- DESIGN.md section 9."""
