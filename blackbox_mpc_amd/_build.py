"""In-tree build of the HIP library (gfx950).  `python -m blackbox_mpc_amd._build` or
`__graft_entry__.build()`.  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbbmpc.so")
SOURCES = ["bbmpc.hip", "bbmpc_fused.hip", "bbmpc_cma.hip", "bbmpc_mlp.hip"]   # engine + ABI | persistent pendulum kernels | CMA-ES | learned-model rollouts
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
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


OBJ_DIR = os.path.join(CSRC, "_obj")
# headers whose kernels are compiled in ONE translation unit only (the text below the marker is preprocessed away
# elsewhere): an edit there rebuilds that unit alone
GUARDED = {"kernels_cma.hpp": ("bbmpc_cma.hip", "#ifdef BBMPC_TU_CMA"),
           "kernels_eigh.hpp": ("bbmpc_cma.hip", "#ifdef BBMPC_TU_CMA"),
           "kernels_eigh_small.hpp": ("bbmpc_cma.hip", "#ifdef BBMPC_TU_CMA"),
           "kernels_fused_cma.hpp": ("bbmpc_cma.hip", "#ifdef BBMPC_TU_CMA")}


def _after_matching_endif(rest):
    """The text behind the #endif that closes a guard whose #ifdef line has just been consumed: conditionals are counted,
    so an include guard around the whole header (or any #if inside the guarded part) does not move the cut."""
    import re
    depth, pos = 1, 0
    for m in re.finditer(r"^[ \t]*#[ \t]*(if|ifdef|ifndef|endif)\b[^\n]*$", rest, flags=re.M):
        depth += -1 if m.group(1) == "endif" else 1
        pos = m.end()
        if depth == 0:
            return rest[pos:]
    return ""


def _fingerprint(src):
    """What translation unit `src` is compiled from: its own text, every header (for a guarded header only the part the
    unit sees), the public header, the flags."""
    import hashlib
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    for path in [os.path.join(CSRC, src), os.path.join(HERE, "..", "include", "bbmpc.h")]:
        h.update(open(path, "rb").read())
    for f in sorted(os.listdir(CSRC)):
        if not (f.endswith(".hpp") or f.endswith(".inc")):
            continue
        text = open(os.path.join(CSRC, f)).read()
        if f in GUARDED and GUARDED[f][0] != src and GUARDED[f][1] in text:
            # the other units see the text in front of the guard AND whatever follows its #endif (by convention only the
            # namespace's closing brace): both are hashed, so code placed behind the guard still rebuilds everybody
            head, rest = text.split(GUARDED[f][1], 1)
            text = head + "\n/*guarded*/\n" + _after_matching_endif(rest)
        h.update(f.encode())
        h.update(text.encode())
    return h.hexdigest()


def _stale_units():
    out = []
    for src in SOURCES:
        obj, stamp = os.path.join(OBJ_DIR, src + ".o"), os.path.join(OBJ_DIR, src + ".sha")
        if not (os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read().strip() == _fingerprint(src)):
            out.append(src)
    return out


LIB_STAMP = LIB + ".sha"       # fingerprint of every unit the library was linked from: travels with the .so (the objects do not)


LINK_FLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC"]
_TOOLCHAIN = None


def _toolchain():
    """`hipcc --version` (a library linked by another ROCm must not be taken for current); None where there is no hipcc --
    a box that cannot build can only take the library it was shipped."""
    global _TOOLCHAIN
    if _TOOLCHAIN is None:
        try:
            out = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True).stdout
            # the version lines only (install paths may differ between boxes of one image)
            _TOOLCHAIN = "\n".join(ln.strip() for ln in out.splitlines() if "version" in ln.lower()) or "unknown"
        except Exception:                                  # noqa: BLE001
            _TOOLCHAIN = ""
    return _TOOLCHAIN or None


def _lib_fingerprint():
    """(sources, toolchain): every translation unit's fingerprint + the link line + the build lists of this file, and the toolchain's version."""
    import hashlib
    h = hashlib.sha256("".join(_fingerprint(s) for s in SOURCES).encode())
    # what of this file decides the output (an edit to the build logic alone must not relink -- on the GPU box that is a
    # full compile, the object cache does not travel)
    h.update(repr((SOURCES, FLAGS, LINK_FLAGS, sorted(GUARDED.items()), EMBEDDED_HEADERS)).encode())
    tc = _toolchain()
    return h.hexdigest(), (hashlib.sha256(tc.encode()).hexdigest()[:16] if tc else None)


def needs_build():
    """False when libbbmpc.so was linked from exactly the sources in the tree, by this link line and toolchain (on the GPU
    box: the .so and its stamp arrive with the snapshot, the object cache does not -- nothing is compiled there; the image,
    hence the toolchain, is the same).  Without a hipcc on the box only the sources are compared."""
    if not (os.path.exists(LIB) and os.path.exists(LIB_STAMP)):
        return True
    stamp = open(LIB_STAMP).read().strip().split("|")
    src, tc = _lib_fingerprint()
    return stamp[0] != src or (tc is not None and len(stamp) > 1 and stamp[1] != tc)


EMBED = os.path.join(CSRC, "_embed.inc")
EMBEDDED_HEADERS = ["fastmath.hpp", "models.hpp"]


def write_embedded_sources():
    """csrc/_embed.inc: the text of the leaf-math headers as C++ raw strings, handed to hiprtc as named headers when a
    user-supplied device function is compiled into a fused rollout kernel next to the built-in model / rewards
    (csrc/rtc.hpp).  System includes are dropped (hiprtc provides the HIP builtins itself).  Rewritten only when the
    text changes, so that it does not retrigger the build."""
    import re
    out = ["// generated by blackbox_mpc_amd/_build.py from %s -- do not edit" % ", ".join(EMBEDDED_HEADERS)]
    for h in EMBEDDED_HEADERS:
        text = open(os.path.join(CSRC, h)).read()
        text = re.sub(r"^\s*#include\s*<[^>]+>\s*$", "", text, flags=re.M)
        assert ')BBSRC"' not in text
        out.append('static const char* const k_embed_%s = R"BBSRC(%s)BBSRC";' % (h.split(".")[0], text))
    new = "\n".join(out) + "\n"
    old = open(EMBED).read() if os.path.exists(EMBED) else None
    if new != old:
        tmp = "%s.tmp.%d" % (EMBED, os.getpid())      # several ranks may call build() at once: never a half-written include
        with open(tmp, "w") as f:
            f.write(new)
        os.replace(tmp, EMBED)


def build(force=False, verbose=False):
    """Compile the stale translation units side by side (objects are cached under csrc/_obj, so a kernel edit costs the
    units that see it: 12-62 s instead of the whole library), link."""
    write_embedded_sources()
    os.makedirs(OBJ_DIR, exist_ok=True)
    if not force and not needs_build():
        return LIB
    stale = list(SOURCES) if force else _stale_units()
    procs = []
    for src in stale:
        obj = os.path.join(OBJ_DIR, src + ".o")
        tmpo = "%s.tmp.%d" % (obj, os.getpid())            # several ranks may call build() at once
        cmd = [_hipcc()] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", tmpo]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, obj, tmpo, _fingerprint(src), subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = []
    for src, obj, tmpo, fp, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            failed.append(out)
            if os.path.exists(tmpo):
                os.remove(tmpo)
            continue
        os.replace(tmpo, obj)
        with open(os.path.join(OBJ_DIR, src + ".sha"), "w") as f:
            f.write(fp)
    if failed:
        raise RuntimeError("hipcc failed:\n" + "\n".join(failed))
    tmp = "%s.tmp.%d" % (LIB, os.getpid())
    link = [_hipcc()] + LINK_FLAGS + [os.path.join(OBJ_DIR, s_ + ".o") for s_ in SOURCES] + ["-o", tmp]
    if verbose:
        print(" ".join(link), file=sys.stderr)
    res = subprocess.run(link, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("link failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, LIB)
    with open(LIB_STAMP, "w") as f:
        src_fp, tc_fp = _lib_fingerprint()
        f.write(src_fp + "|" + (tc_fp or ""))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
