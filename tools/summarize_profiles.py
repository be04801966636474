"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into small text summaries for profiles/."""
import collections
import csv
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
dst = os.path.join("gpurun_out", "profiles_" + tag)
os.makedirs(dst, exist_ok=True)


def short(name):
    return name.replace("bbmpc::", "").split("(")[0][:60]


traffic = {}
sq = {}
for cfg_dir in sorted(glob.glob(os.path.join(src, "cfg*"))):
    if not os.path.isdir(cfg_dir):
        continue
    cfg = os.path.basename(cfg_dir)
    lines = ["# %s -- rocprofv3 summary (%s)" % (cfg, tag), ""]
    bj = os.path.join(src, cfg + ".bench.json")
    if os.path.exists(bj):
        try:
            lines += ["bench line under the profiler (kernel trace on):", "```", open(bj).read().strip(), "```", ""]
        except Exception:
            pass
    st = glob.glob(os.path.join(cfg_dir, "*trace_kernel_stats.csv"))
    avg_ns = {}
    if st:
        for r in csv.DictReader(open(st[0])):
            avg_ns[short(r["Name"])] = float(r["AverageNs"])
        lines += ["## kernel stats (rocprofv3 --kernel-trace --stats)", "", "| kernel | calls | total ns | avg ns | % |",
                  "|---|---|---|---|---|"]
        for r in csv.DictReader(open(st[0])):
            lines.append("| %s | %s | %s | %s | %s |" % (short(r["Name"]), r["Calls"], r["TotalDurationNs"],
                                                        r["AverageNs"], r["Percentage"]))
        lines.append("")
    pm = {}
    for kind in ("fetch", "write", "sq"):
        f = glob.glob(os.path.join(cfg_dir, "*%s_counter_collection.csv" % kind))
        if not f:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.defaultdict(collections.Counter)
        for r in csv.DictReader(open(f[0])):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
        for k in agg:
            for c in agg[k]:
                pm.setdefault(k, {})[c] = agg[k][c] / cnt[k][c]
    if pm:
        lines += ["## PMC counters, mean per dispatch (separate passes: FETCH_SIZE | WRITE_SIZE | SQ_*)", ""]
        for k, d in sorted(pm.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
            if not any(x in k for x in ("k_", "rollout", "fused")):
                continue
            lines.append("* `%s`: " % k + ", ".join("%s=%.4g" % (c, v) for c, v in sorted(d.items())))
            if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
                # MI355X_MICROARCH.md: counters are in KiB; FETCH_SIZE under-reports wide streaming reads by 2x
                hbm = (2.0 * d.get("FETCH_SIZE", 0.0) + d.get("WRITE_SIZE", 0.0)) * 1024.0
                lines.append("  * HBM traffic per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 = %.0f bytes" % hbm)
                traffic.setdefault(cfg, {})[k] = hbm
            if "SQ_INSTS_VALU" in d:
                sq.setdefault(cfg, {})[k] = {c: d[c] for c in d if c.startswith("SQ_")}
            if d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0 and k in avg_ns:
                # matrix-pipe utilisation: busy cycles summed over every SIMD of the part / (kernel duration x clock x
                # 256 CUs x 4 SIMDs); 2.13 GHz is the clock measured under matrix load, 2.4 GHz the peak
                dur = avg_ns[k] * 1e-9
                for clk in (2.13e9, 2.4e9):
                    lines.append("  * MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (%.2f us x %.2f GHz x 256 CUs x 4 SIMDs) = %.3f"
                                 % (avg_ns[k] * 1e-3, clk * 1e-9, d["SQ_VALU_MFMA_BUSY_CYCLES"] / (dur * clk * 256 * 4)))
        lines.append("")
    open(os.path.join(dst, "%s_%s.md" % (tag, cfg)), "w").write("\n".join(lines) + "\n")
# the shape every configuration was profiled at (bench.py attaches counters only to a run of the same shape)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
try:
    import bench as _bench
    shapes = {c: {k: _bench.CONFIGS[c].get(k) for k in _bench.SHAPE_KEYS} for c in set(traffic) | set(sq) if c in _bench.CONFIGS}
    traffic["_shapes"] = shapes
    sq["_shapes"] = dict(shapes)
except Exception as exc:
    print("summarize_profiles: shapes not recorded (%s)" % exc)
json.dump(traffic, open(os.path.join(dst, "%s_hbm_traffic.json" % tag), "w"), indent=1, sort_keys=True)
json.dump(sq, open(os.path.join(dst, "%s_sq_counters.json" % tag), "w"), indent=1, sort_keys=True)
print("wrote", os.listdir(dst))
