"""Merge a partial re-profile (gpurun_out/profiles_<tag>, gpurun_out/prof_<tag>) into the tracked profiles/:
per-configuration .md and kernel-stats CSV are replaced, the per-configuration entries of <tag>_hbm_traffic.json and
<tag>_sq_counters.json are replaced, the other configurations keep their entries.
usage: python tools/merge_profiles.py r3"""
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
src = os.path.join("gpurun_out", "profiles_" + tag)
raw = os.path.join("gpurun_out", "prof_" + tag)
for md in glob.glob(os.path.join(src, "%s_cfg*.md" % tag)):
    shutil.copy(md, "profiles/")
    cfg = os.path.basename(md)[len(tag) + 1:-3]
    st = glob.glob(os.path.join(raw, cfg, "*trace_kernel_stats.csv"))
    if st:
        shutil.copy(st[0], os.path.join("profiles", "%s_%s_kernel_stats.csv" % (tag, cfg)))
    print("merged", cfg)
for name in ("hbm_traffic", "sq_counters"):
    new = os.path.join(src, "%s_%s.json" % (tag, name))
    old = os.path.join("profiles", "%s_%s.json" % (tag, name))
    if not os.path.exists(new):
        continue
    d = json.load(open(old)) if os.path.exists(old) else {}
    nd = json.load(open(new))
    shapes = dict(d.get("_shapes", {}))
    shapes.update(nd.get("_shapes", {}))          # (configurations that were not re-profiled keep their recorded shape)
    d.update(nd)
    if shapes:
        d["_shapes"] = shapes
    json.dump(d, open(old, "w"), indent=1, sort_keys=True)
