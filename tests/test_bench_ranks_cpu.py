"""bench.py's rank logic on CPU: the launch line the driver uses for N > 1 (`python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N ...`), under BBMPC_BENCH_BACKEND=gloo with the stub workload (no GPU here).
What is under test: RANK / LOCAL_RANK / WORLD_SIZE handling, process-group set-up and tear-down, the barrier-bracketed
timing with max over ranks, the gathered-rows check, and that rank 0 prints exactly one JSON line with the contract's
keys -- so the 8-rank path cannot rot between GPU runs."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(nproc, extra_env=None, args=()):
    env = dict(os.environ, BBMPC_BENCH_BACKEND="gloo", BBMPC_BENCH_STUB="1", OMP_NUM_THREADS="1")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", str(nproc), "--steps", "12", "--warmup", "3", "--no-cpu-baseline"] + list(args)
    res = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 must print exactly one JSON line, got: %r" % res.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("nproc,config", [(2, "cfg3"), (3, "cfg5pso")])
def test_multi_rank_launch_line(nproc, config):
    out = _run(nproc, args=("--config", config))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "median_ms", "p10_ms", "p90_ms",
                "device_resident_control_steps_per_sec", "multi_gpu"):
        assert key in out, key
    assert out["n_gpus"] == nproc and out["steps"] == 12 and out["warmup"] == 3 and out["scaling"] == "weak"
    assert out["percentile_samples"] == 30                       # >= 30 samples for the median even when --steps is smaller
    mg = out["multi_gpu"]
    assert mg["ranks"] == nproc and mg["fallback_reason"] is None and mg["gather_mode"].startswith("torch.distributed")
    # 2 slots x world*A rows, checked after the act region and after the device-resident region
    import bench
    assert mg["gathered_rows_checked"] == 2 * 2 * nproc * bench.CONFIGS[config]["A"]
    assert out["value"] > 0 and out["p10_ms"] <= out["median_ms"] <= out["p90_ms"]


def test_plain_launch_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (what a driver that does not wrap the call would run): the
    script re-runs itself under torch.distributed.run and rank 0's line comes out of the parent's stdout.  The default
    configuration (BASELINE configs[1], weak scaling) carries BASELINE configs[2] -- 64 agents in total, 64 / N per GPU --
    as its `secondary`, and both say how their exchange ran."""
    env = dict(os.environ, BBMPC_BENCH_BACKEND="gloo", BBMPC_BENCH_STUB="1", OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "3",
                          "--no-cpu-baseline"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and "started its own ranks" in out["launch"]
    assert out["multi_gpu"]["ranks"] == 2 and out["multi_gpu"]["gathered_rows_checked"] > 0
    for key in ("value_launch_per_call", "resident", "launch_per_call_median_ms"):
        assert key in out, key
    sec = out["secondary"]
    assert sec["scaling"] == "strong" and sec["n_gpus"] == 2 and "64 agents in total, 32 per GPU" in sec["note"]
    assert sec["multi_gpu"]["ranks"] == 2 and sec["multi_gpu"]["gathered_rows_checked"] == 2 * 2 * 64
    assert "PI2" in sec["metric"] and sec["value"] > 0


def test_world_size_mismatch_is_an_error_message():
    env = dict(os.environ, BBMPC_BENCH_BACKEND="gloo", BBMPC_BENCH_STUB="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"], capture_output=True,
                         text=True, env=env, cwd=ROOT, timeout=120)
    assert res.returncode != 0 and "--gpus 2 but the launcher started 1 rank" in (res.stdout + res.stderr)


def test_stub_is_refused_outside_gloo():
    env = dict(os.environ, BBMPC_BENCH_STUB="1", BBMPC_BENCH_BACKEND="nccl")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2"], capture_output=True, text=True,
                         env=env, cwd=ROOT, timeout=120)
    assert res.returncode != 0 and "BBMPC_BENCH_STUB" in (res.stdout + res.stderr)
