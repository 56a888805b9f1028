#!/usr/bin/env python3
"""Dev tool: a build of libqa_sim.so with extra -D switches on qa_sim.hip (the env kernels), as tools/_prof/libqa_sim_<name>.so, for A/B timing
on the GPU box (QA_LIB=<path> python tools/quick_time.py).  The other translation units are compiled once into tools/_prof/obj/.

  python tools/build_variant.py NAME [-DFLAG ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g      # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
prof = os.path.join(ROOT, "tools", "_prof"); objd = os.path.join(prof, "obj"); os.makedirs(objd, exist_ok=True)
base = [g._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
objs = []
for src, extra in g.HIP_SOURCES:
    path = os.path.join(g.CSRC, src)
    if src == "qa_sim.hip":
        obj = os.path.join(objd, f"qa_sim_{name}.o")
        subprocess.check_call(base + extra + flags + ["-c", path, "-o", obj])
    else:
        obj = os.path.join(objd, src.replace(".hip", ".o"))
        if not os.path.exists(obj) or os.path.getmtime(obj) < os.path.getmtime(path):
            subprocess.check_call(base + extra + ["-c", path, "-o", obj])
    objs.append(obj)
out = os.path.join(prof, f"libqa_sim_{name}.so")
subprocess.check_call([g._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print(out)
