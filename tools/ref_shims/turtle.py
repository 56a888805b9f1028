"""bbc/rsl_rl/modules/estimator.py:1 has a stray `from turtle import forward` (needs tkinter)."""


def forward(*a, **k):
    raise RuntimeError("turtle stub")
