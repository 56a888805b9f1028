"""quadrupedal-agility_amd: MI355X-native drop-in for the legged_gym step() + rsl_rl PPO/AMP hot
path of NJU-RLC/quadrupedal-agility.  The compute lives in csrc/ (HIP, gfx950) behind the C ABI
of include/qa_sim.h; `legged_gym` / `rsl_rl` below mirror the reference's host-side interface.
"""
import importlib
import sys

__version__ = "0.1.0"


def install_reference_aliases():
    """Make `import legged_gym` / `import rsl_rl` resolve to this package's mirrors, so that
    reference-style scripts and pickled `rsl_rl.utils.utils.Normalizer` objects inside model.pt
    (on_policy_runner.py:306-321) load unchanged."""
    for short in ("legged_gym", "rsl_rl"):
        if short not in sys.modules:
            sys.modules[short] = importlib.import_module(f"{__name__}.{short}")
    for sub in ("rsl_rl.utils", "rsl_rl.utils.utils"):
        if sub not in sys.modules:
            sys.modules[sub] = importlib.import_module(f"{__name__}.{sub}")
