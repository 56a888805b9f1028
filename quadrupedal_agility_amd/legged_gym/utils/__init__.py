from .helpers import class_to_dict, get_args, get_load_path, set_seed, update_cfg_from_args  # noqa: F401
from .task_registry import task_registry  # noqa: F401
