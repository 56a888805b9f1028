"""Only reached by the reference's non-batched mocap paths, which the golden generator never calls."""


def quaternion_slerp(*a, **k):
    raise NotImplementedError


def quaternion_about_axis(*a, **k):
    raise NotImplementedError
