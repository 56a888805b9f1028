"""Nested-class configuration base (API of bbc/legged_gym/envs/base/base_config.py:4-26).

Instantiating a config turns every nested class attribute into an instance, recursively, so
`cfg.env.num_envs = 64` edits that instance and not the shared class.
"""
import inspect


class BaseConfig:
    def __init__(self):
        _instantiate_nested(self)


def _instantiate_nested(node):
    for name in dir(node):
        if name == "__class__":
            continue
        member = getattr(node, name)
        if inspect.isclass(member):
            inst = member()
            setattr(node, name, inst)
            _instantiate_nested(inst)
