from .on_policy_runner import OnPolicyRunner  # noqa: F401
