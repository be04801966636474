"""In-tree build of the HIP library (gfx950).  `python -m blackbox_mpc_amd._build` or
`__graft_entry__.build()`.  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbbmpc.so")
SOURCES = ["bbmpc.hip"]
HEADERS = ["engine.hpp", "kernels_rollout.hpp", "kernels_refit.hpp", "models.hpp", "rng.hpp", "kernels_mlp.hpp", "kernels_fused.hpp", "fastmath.hpp", "topk.hpp",
           "kernels_opt.hpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         # one rounding per reference op: never contract a*b+c behind the source's back
         "-ffp-contract=off",
         # no SLP vectorisation: on gfx950 a packed v_pk_mul/add_f32 takes two issue slots (tools/microbench/pk_fp32.hip),
         # so pairing scalar fp32 ops buys nothing and costs the v_movs that assemble the register pairs
         "-fno-slp-vectorize",
         "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=...)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    # every file under csrc/ (headers are all included by the one translation unit), the public header, and this
    # file itself (the compiler flags live here)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "bbmpc.h"),
                                                                os.path.abspath(__file__)]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    tmp = "%s.tmp.%d" % (LIB, os.getpid())       # several ranks may call build() at once
    cmd = [_hipcc()] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", tmp]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
