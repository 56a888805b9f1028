"""TSC (task-level controller) tree of the reference.  Learner side: the hybrid categorical + Gaussian
policy `ActorCriticTSC`, the frozen low-level `ActorCriticBBC`, the hybrid `PPO` and its rollout storage (`tsc.rsl_rl`).
Env side: the command mapping and the goal / termination / reward bookkeeping of the task-level `LeggedRobot`
(`tsc.legged_gym.TaskLevelBookkeeping`, HIP kernels `qa_tsc_set_commands` / `qa_tsc_goal_step`).  The obstacle-course
simulation itself (obstacle contact, see-saw joints, the 132-point scan, depth camera) is NOT built -- DESIGN.md section 9."""
