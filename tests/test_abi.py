"""CPU-side checks of the C-ABI boundary: the HIP library loads without a GPU and exports every symbol that
include/qa_sim.h declares; the oracle exports the same set under qo_; both agree on the arena layout."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from quadrupedal_agility_amd import _capi
from tests.oracle_lib import go2_cfg, load_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "qa_sim.h")).read()
    names = set(re.findall(r"\b(qa_[a-z_]+)\s*\(", text))
    return sorted(n for n in names)


@pytest.fixture(scope="module")
def hip_lib():
    import __graft_entry__ as g
    g.build()
    return _capi.load_library()


def test_header_symbols_exported(hip_lib):
    declared = _declared_symbols()
    assert set("qa_" + s for s in _capi.ABI_SYMBOLS) == set(declared)
    for name in declared:
        assert hasattr(hip_lib, name), f"{name} declared in include/qa_sim.h but not exported by libqa_sim.so"


def test_oracle_exports_twin_symbols():
    lib = load_oracle()
    for s in _capi.ABI_SYMBOLS:
        if s in ("last_error", "abi_version"):
            continue
        assert hasattr(lib, "qo_" + s)


def test_layout_agrees_between_library_and_oracle(hip_lib):
    lib_o = load_oracle()
    for n in (1, 17, 4096):
        q = go2_cfg(n)
        assert hip_lib.qa_arena_bytes(C.byref(q)) == lib_o.qo_arena_bytes(C.byref(q))
        for name, idx in _capi.T.items():
            assert _capi.tensor_info(hip_lib, "qa_", q, idx) == _capi.tensor_info(lib_o, "qo_", q, idx), name
    q = go2_cfg(33); q.terrain_type = 1; q.hf_rows, q.hf_cols, q.hf_hscale, q.hf_vscale, q.hf_border = 321, 123, 0.1, 0.005, 2.0
    assert hip_lib.qa_arena_bytes(C.byref(q)) == lib_o.qo_arena_bytes(C.byref(q))
    for name, idx in _capi.T.items():
        assert _capi.tensor_info(hip_lib, "qa_", q, idx) == _capi.tensor_info(lib_o, "qo_", q, idx), name
    assert _capi.tensor_info(hip_lib, "qa_", q, _capi.T["HEIGHT_SAMPLES"])[1:] == ((321, 123), _capi.DTYPE_I16)
    q = go2_cfg(4096)
    off, shape, dt = _capi.tensor_info(hip_lib, "qa_", q, _capi.T["OBS"])
    assert shape == (4096, 671) and dt == _capi.DTYPE_F32 and off % 256 == 0


def test_config_struct_size_matches_c():
    # the oracle is plain C: ask it how big it thinks qa_config is by probing an out-of-range read guard
    q = go2_cfg(8)
    assert C.sizeof(q) == 608          # ABI v11: + articulated_obstacles, reserved0
    lib = load_oracle()
    assert lib.qo_arena_bytes(C.byref(q)) > 0
    assert q.max_episode_length == 1000 and q.resampling_steps == 300 and q.push_interval == 400


def test_bad_arguments_return_error_codes(hip_lib):
    q = go2_cfg(8)
    assert hip_lib.qa_arena_bytes(None) == -1
    h = C.c_void_p()
    assert hip_lib.qa_create(C.byref(q), None, 0, None, C.byref(h)) == -1
    q.abi_version = 99
    buf = np.zeros(1 << 20, np.uint8)
    assert hip_lib.qa_create(C.byref(q), buf.ctypes.data, buf.nbytes, None, C.byref(h)) == -4


def test_product_has_no_cpu_fallback():
    """The package must not import or reference the oracle."""
    pkg = os.path.join(ROOT, "quadrupedal_agility_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dp, f)).read()
                assert "libqa_oracle" not in text and "qo_" not in text.replace("qo_stats", ""), os.path.join(dp, f)


def test_library_builds_from_scratch(tmp_path):
    """`build()` reuses the shipped library whenever its source hash matches, so nothing showed that the sources COMPILE on
    the image they are judged on.  This compiles every HIP source for gfx950 from scratch (hipcc cross-compiles without a GPU; ~40 s) into a
    scratch directory -- the same `compile_library` that `QA_FORCE_REBUILD=1 python __graft_entry__.py` runs in place -- and checks that the
    result exports the whole ABI."""
    import __graft_entry__ as g
    out = str(tmp_path / "libqa_sim_scratch.so")
    g.compile_library(out, obj_dir=str(tmp_path))
    lib = C.CDLL(out)
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    lib.qa_abi_version.restype = C.c_int
    assert lib.qa_abi_version() == _capi.QA_ABI_VERSION
