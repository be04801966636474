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
