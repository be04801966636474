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


def test_stub_is_refused_outside_gloo():
    env = dict(os.environ, BBMPC_BENCH_STUB="1", BBMPC_BENCH_BACKEND="nccl")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2"], capture_output=True, text=True,
                         env=env, cwd=ROOT, timeout=120)
    assert res.returncode != 0 and "BBMPC_BENCH_STUB" in (res.stdout + res.stderr)
