from .motion_loader import MotionLoader  # noqa: F401
