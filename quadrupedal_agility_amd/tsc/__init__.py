"""TSC (task-level controller) tree of the reference, learner side only so far: the hybrid
categorical + Gaussian policy `ActorCriticTSC`, the frozen low-level `ActorCriticBBC`, the hybrid `PPO` and its rollout
storage.  The TSC environment (obstacle course, goals, two-level stepping) is NOT built yet -- DESIGN.md section 9."""
