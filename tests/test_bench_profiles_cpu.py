"""bench.py attaches rocprofv3 counters (HBM traffic, VALU instructions) to a roofline block only when the committed
profile was taken at the shape that is being timed (the config-3 block looked its counters
up by name and got the 8-agent profile for a 64-agent run)."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_profile_lookup_goes_by_shape_not_by_name():
    c3 = dict(bench.CONFIGS["cfg3"])
    assert bench.profile_config_for(c3) == "cfg3"
    assert bench.profile_config_for(dict(c3, A=64)) == "cfg3full"        # run_block("cfg3", agents=64): BASELINE configs[2] on one GPU
    assert bench.profile_config_for(dict(c3, A=16)) is None              # 4-GPU share: no profile of that shape -> nothing attached
    assert bench.profile_config_for(dict(bench.CONFIGS["cfg2"])) == "cfg2"


def test_committed_profiles_record_matching_shapes():
    for kind in ("hbm_traffic", "sq_counters"):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_%s.json" % kind)))
        assert files, "no committed %s profile" % kind
        d = json.load(open(files[-1]))
        for name in d:
            if name.startswith("_"):
                continue
            assert name in bench.CONFIGS, name
            assert bench.profile_shape_matches(d, name, bench.CONFIGS[name]), (files[-1], name)
            # a run of the same configuration with another agent count must NOT match
            other = dict(bench.CONFIGS[name], A=bench.CONFIGS[name]["A"] + 1)
            assert not bench.profile_shape_matches(d, name, other)


# ---- the counters of a roofline block belong to the INSTANTIATION that was timed (the driver line of
# round 4 divided the strict-math instantiation's SQ_INSTS_VALU by the default instantiation's time: frac 1.02 / 1.50) ----
class _W:
    """What bench.roofline needs of a workload."""

    def __init__(self, name, agents=None):
        self.c = dict(bench.CONFIGS[name])
        if agents is not None:
            self.c["A"] = agents
        self.mlp = self.c["env"] == "cheetah"
        self.U, self.S = (6, 20) if self.mlp else (1, 3)


def _stub_measurement(kname, kinst, avg_us):
    return dict(roll_ms=avg_us * 1e-3 * 10, roll_n=10, kname=kname, kinst=kinst, prof_every=1)


def _latest(kind):
    return json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "*_%s.json" % kind)))[-1]))


def test_profile_entry_needs_the_exact_instantiation():
    ent = {"void k_fused_pendulum<2, true, false, 2, 1, false>": 1, "void k_fused_pendulum<2, true, true, 2, 1, false>": 2,
           "void k_fused_pendulum<2, true, true, 2, 1, true>": 3, "k_noise_fill": 4, "void k_rollout_mlp_q4s<50, 7, 1, 1, 0, 1>": 5}
    pe = bench.profile_entry
    assert pe(ent, "k_fused_pendulum", "k_fused_pendulum<2, true, true, 2, 1, false>")[1] == 2
    assert pe(ent, "k_fused_pendulum", "k_fused_pendulum<2, true, false, 2, 1, false>")[1] == 1
    assert pe(ent, "k_fused_pendulum", "k_fused_pendulum<2, true, true, 2, 1, true>")[1] == 3
    assert pe(ent, "k_fused_pendulum", "k_fused_pendulum") is None                      # ambiguous plain name: attach nothing
    assert pe(ent, "k_fused_pendulum", "k_fused_pendulum<3, true, true, 2, 1, false>") is None
    assert pe(ent, "k_rollout_mlp_q4s", "k_rollout_mlp_q4s")[1] == 5                  # one instantiation: the plain name will do
    assert pe(ent, "k_noise_fill", "k_noise_fill")[1] == 4
    assert pe(ent, "k_rollout_mlp", "k_rollout_mlp") is None                            # not a prefix match


def test_roofline_counters_follow_the_timed_instantiation_and_stay_below_one():
    """Stub measurements at the durations of the committed kernel statistics: every fraction <= 1, and the counter's kernel
    carries the FASTM flag of the run (template argument 3 of k_fused_pendulum)."""
    sq, tr = _latest("sq_counters"), _latest("hbm_traffic")
    import csv
    for cfg, agents in (("cfg2", None), ("cfg3", None), ("cfg3", 64)):
        W = _W(cfg, agents)
        prof = bench.profile_config_for(W.c)
        stats = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_%s_kernel_stats.csv" % prof)))[-1]
        # (the statistics file spells "void bbmpc::k<...>(bbmpc::FusedArgs)", the counter tables "void k<...>")
        dur = {r["Name"].replace("bbmpc::", "").split("(")[0].strip(): float(r["AverageNs"]) * 1e-3 for r in csv.DictReader(open(stats))}
        for kn in sq[prof]:
            if "k_fused_pendulum" not in kn or kn.rstrip().endswith("true>"):
                continue             # (the resident instantiation serves many control steps per dispatch: no per-launch figure)
            inst = kn.replace("void ", "").strip()
            fastm = inst.split(",")[2].strip()
            r = bench.roofline(W, _stub_measurement("k_fused_pendulum", inst, dur[kn]), cfg, 1)
            assert r["kernel_instantiation"] == inst
            v = r["valu_issue"]
            assert v["kernel"].replace("void ", "").strip() == inst and v["kernel"].split(",")[2].strip() == fastm
            assert v["insts_per_launch"] == sq[prof][kn]["SQ_INSTS_VALU"]
            assert 0.0 < v["frac_guide"] < v["frac"] <= 1.0, (cfg, inst, v)
            assert r["traffic"] == tr[prof][kn]
            assert 0.0 < r["frac"] <= 1.0
    # an instantiation the profile does not hold, or the bare name: nothing attached rather than somebody else's counter
    W = _W("cfg2")
    for inst in ("k_fused_pendulum", "k_fused_pendulum<2, false, true, 0, 2, false>"):
        r = bench.roofline(W, _stub_measurement("k_fused_pendulum", inst, 46.0), "cfg2", 1)
        assert r["traffic"] is None and "valu_issue" not in r


def test_round4_driver_line_defect_is_what_this_guards():
    """The strict-math counter over the default kernel's time is > 1 of the measured peak: the figure BENCH_r04 carried."""
    sq = _latest("sq_counters")["cfg2"]
    strict = [v for k, v in sq.items() if "k_fused_pendulum<2, true, false" in k]
    if not strict:
        return
    peak = 4 / bench.VALU_ISSUE_NS_MEASURED * 1e9
    assert strict[0]["SQ_INSTS_VALU"] / 46.1e-6 / peak > 1.0
