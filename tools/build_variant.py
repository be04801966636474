#!/usr/bin/env python
"""A/B builds of ONE translation unit: tools/variants/libbbmpc_<name>.so = the cached objects of the tree with `unit`
recompiled under extra -D flags.  Run the benchmark against it with BBMPC_LIB=tools/variants/libbbmpc_<name>.so (the
variants are git-ignored and travel to the GPU box with the snapshot).

    python tools/build_variant.py r4pair bbmpc_mlp.hip -DBBMPC_PAIR_OPTS=0
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from blackbox_mpc_amd import _build as B  # noqa: E402


def main():
    name, unit, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    B.build()                                            # the cached objects of every other unit
    out_dir = os.path.join(ROOT, "tools", "variants")
    os.makedirs(out_dir, exist_ok=True)
    obj = os.path.join(out_dir, "%s.%s.o" % (name, unit))
    subprocess.check_call([B._hipcc()] + B.FLAGS + extra + ["-c", os.path.join(B.CSRC, unit), "-o", obj])
    objs = [obj if s == unit else os.path.join(B.OBJ_DIR, s + ".o") for s in B.SOURCES]
    lib = os.path.join(out_dir, "libbbmpc_%s.so" % name)
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib])
    os.remove(obj)
    print(lib)


if __name__ == "__main__":
    main()
