"""Learner-side mirror of the reference's `rsl_rl` package (BBC tree): same public classes and
checkpoint layout, host syncs removed from the hot loop, GAE on the HIP kernel."""
