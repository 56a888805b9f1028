"""Env-side pieces of the task-level tree that need no obstacle physics."""
from .task_level import TaskLevelBookkeeping

__all__ = ["TaskLevelBookkeeping"]
