from . import transformations  # noqa: F401
