"""Two real ranks, one GPU each, over RCCL: skipped on a one-GPU box, runs by itself the day the
`-m gpu` tier gets a node with two devices.  The one-rank forms of the same paths are tests/test_gpu_comm.py and
tests/test_gpu_popshard.py; the rank logic under gloo is tests/test_parallel_cpu.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _two_gpus():
    from blackbox_mpc_amd import _build
    _build.build()
    from blackbox_mpc_amd import _lib
    return _lib.device_count() >= 2


def _env():
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"          # the host driver only supports dmabuf IPC (RCCL across processes)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    return env


def test_two_ranks_agent_shards_and_population_shards(tmp_path):
    if not _two_gpus():
        pytest.skip("needs two GPUs")
    env = _env()
    env["BBMPC_TWO_RANK_OUT"] = str(tmp_path / "two")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29741", os.path.join(ROOT, "tests", "two_rank_worker.py")]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    for r in range(2):
        out = json.load(open("%s.rank%d.json" % (env["BBMPC_TWO_RANK_OUT"], r)))
        assert out["rccl_ranks"] == 2
        # (i) the gathered records are the unsharded engine's, bit for bit, on every rank
        assert out["agent_shard_gather_max_abs_diff"] == 0.0
        # (ii) population shards against the unsharded optimizer: order of the fp32 sums
        assert out["popshard_max_abs_diff"]["PI2"] <= 2e-5
        assert out["popshard_max_abs_diff"]["CEM"] <= 2e-5


def test_worker_script_with_one_rank(tmp_path):
    # the same script as a one-rank group on one GPU: keeps it runnable until a two-GPU box shows up
    env = _env()
    env["BBMPC_TWO_RANK_OUT"] = str(tmp_path / "one")
    env["BBMPC_TWO_RANK_ALLOW_ONE"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29743", os.path.join(ROOT, "tests", "two_rank_worker.py")]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    out = json.load(open("%s.rank0.json" % env["BBMPC_TWO_RANK_OUT"]))
    assert out["rccl_ranks"] == 1 and out["agent_shard_gather_max_abs_diff"] == 0.0
    assert max(out["popshard_max_abs_diff"].values()) <= 2e-5


def test_bench_with_two_gpus_prints_one_line():
    if not _two_gpus():
        pytest.skip("needs two GPUs")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                          "--no-cpu-baseline"], env=_env(), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{\"metric\"")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["multi_gpu"]["rccl_ranks"] == 2
