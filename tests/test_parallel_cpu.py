"""N>1 path on CPU: world_size-2 gloo run of the agent sharding + record gather (the only exchange the hot
path has).  The rank-local 'policy' here is the oracle (no GPU in this container); what is under test is
that sharded == unsharded per agent and that the gather restores global agent order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from blackbox_mpc_amd.parallel import agent_shard


def test_agent_shard_partitions():
    for A in (1, 5, 8, 32, 64):
        for W in (1, 2, 3, 4, 8):
            blocks = [agent_shard(A, W, r) for r in range(W)]
            assert sum(c for _, c in blocks) == A
            pos = 0
            for off, c in blocks:
                assert off == pos
                pos += c
            assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1
    assert agent_shard(64, 8, 3) == (24, 8) and agent_shard(32, 8, 7) == (28, 4)
    with pytest.raises(ValueError):
        agent_shard(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _OraclePolicy:
    """rank-local stand-in for MPCPolicy: RandomSearch through the oracle with noise keyed by GLOBAL agent id"""

    def __init__(self, offset, count, total):
        from oracle import oracle_np as O
        self.O, self.offset, self.count = O, offset, count
        ev = O.Evaluator("pendulum", O.Handler(O.pendulum_dynamics, True))
        self.opt = O.RandomSearch(ev, [-2.0], [2.0], horizon=6, population=32, num_agents=count)

    def reset(self):
        pass

    def act(self, obs, t, exploration_noise=False):
        u01 = np.stack([np.random.default_rng(1000 * t + self.offset + a).random((32, 6, 1)) for a in range(self.count)],
                       axis=1).astype(np.float32)
        return self.opt.call(obs, {"uniform": u01})


def _worker(rank, world, port, A, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from blackbox_mpc_amd.parallel import ShardedMPCPolicy, gather_records
        from oracle import oracle_np as O
        off, cnt = agent_shard(A, world, rank)
        # 1. raw gather: record row = global agent id
        local = torch.arange(off, off + cnt, dtype=torch.float32)[:, None].repeat(1, 5)
        full = gather_records(local, A)
        assert full.shape == (A, 5) and torch.equal(full[:, 0], torch.arange(A, dtype=torch.float32))
        # 2. sharded policy == unsharded policy, per agent, for 3 closed-loop control steps
        pol = ShardedMPCPolicy(_OraclePolicy, A)
        ref = _OraclePolicy(0, A, A)
        obs = O.pendulum_start_states(A)
        for t in range(3):
            a, n, r = pol.act(obs, t)
            a0, n0, r0 = ref.act(obs, t)
            np.testing.assert_array_equal(a, a0)
            np.testing.assert_array_equal(n, n0)
            np.testing.assert_array_equal(r, r0)
            obs = n
        q.put((rank, "ok"))
    except Exception as e:          # surface the failure in the parent
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("A", [4, 5])
def test_sharded_equals_unsharded_gloo_world2(A):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, A, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_sharded_policy_rejects_more_ranks_than_agents():
    # a rank without agents has no record layout: refused at construction, before any collective is joined
    from blackbox_mpc_amd.parallel import ShardedMPCPolicy, population_shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        with pytest.raises(ValueError, match="world_size"):
            ShardedMPCPolicy(lambda off, cnt, tot: None, num_agents_global=0)
    finally:
        dist.destroy_process_group()
    assert population_shard(1000, 4, 3) == (750, 250)
    with pytest.raises(ValueError):
        population_shard(1000, 3, 0)
