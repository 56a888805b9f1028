"""-m gpu: the C ABI driven from a C++ host with no Python in the process (tests/c_host/qa_host_check.cpp): qa_create on a
hipMalloc'ed arena, qa_env_step on a user stream, results read back through qa_tensor_info offsets, checked against the oracle
loaded through the same ABI.  This is the binding a C/C++ engine would write ("C / C++ hosts")."""
import ctypes as C
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_drives_the_library_and_matches_the_oracle(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    from tests.oracle_lib import go2_cfg, load_oracle
    load_oracle()                                             # builds oracle/libqa_oracle.so if needed
    exe = str(tmp_path / "qa_host_check")
    csrc = os.path.join(ROOT, "quadrupedal_agility_amd", "csrc")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "c_host", "qa_host_check.cpp"),
                           "-I" + os.path.join(ROOT, "include"), "-L" + csrc, "-lqa_sim", "-ldl", "-o", exe])
    cfg = go2_cfg(256, seed=3)
    cfg_path = str(tmp_path / "cfg.bin")
    with open(cfg_path, "wb") as f:
        f.write(bytes(cfg))
    assert len(bytes(cfg)) == C.sizeof(cfg)
    env = dict(os.environ, LD_LIBRARY_PATH=csrc + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([exe, cfg_path, os.path.join(ROOT, "oracle", "libqa_oracle.so"), "20"], env=env, capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "c_host:" in out.stdout
