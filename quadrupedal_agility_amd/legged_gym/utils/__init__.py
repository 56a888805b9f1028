from .helpers import class_to_dict, export_policy_as_jit, get_args, get_load_path, set_seed, update_cfg_from_args  # noqa: F401
from .logger import Logger  # noqa: F401


def __getattr__(name):
    # lazy: task_registry pulls in the runner and the env package; importing it eagerly here makes
    # `import quadrupedal_agility_amd.rsl_rl...` circular (the reference has the same knot)
    if name == "task_registry":
        from .task_registry import task_registry
        globals()["task_registry"] = task_registry      # the import above bound the SUBMODULE to this name; rebind the object
        return task_registry
    raise AttributeError(name)
