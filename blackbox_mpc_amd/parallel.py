"""Agent sharding across the GPUs of one node.

Every optimizer reduction in the reference is per agent along the population axis and the evaluator
treats rows independently, so agents shard with no data-path collective: rank r owns a contiguous block
of agents, runs their whole control step locally (RNG keyed by GLOBAL agent id, so sharded == unsharded
bit for bit) and one all-gather of the packed (action | next_state | reward) records per control step is
the only exchange.  One process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm;
"gloo" on CPU for tests)."""
import numpy as np


def agent_shard(num_agents_global, world_size, rank):
    """Contiguous block of agents owned by `rank`: (offset, count).  Sizes differ by at most one."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, extra = divmod(int(num_agents_global), int(world_size))
    count = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return offset, count


def population_shard(population_global, world_size, rank):
    """Population sharding for num_agents < n_gpus (any optimizer): the contiguous block of particles
    (offset, count) rank `rank` rolls out.  Every shard must have the same size (the engine all-gathers fixed-size
    partials and keys its RNG by global particle index), so the population must divide evenly."""
    if population_global % world_size:
        raise ValueError("population_size %d does not divide over %d ranks" % (population_global, world_size))
    count = population_global // world_size
    return rank * count, count


def gather_records(local_record, num_agents_global, group=None):
    """All-gather the per-agent records of every rank into global agent order.

    local_record: torch tensor [A_local, W] (device tensor for nccl, CPU tensor for gloo).
    Returns a tensor [num_agents_global, W] on the same device, identical on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    width = local_record.shape[1]
    counts = [agent_shard(num_agents_global, world, r)[1] for r in range(world)]
    cap = max(counts)
    if all(c == cap for c in counts):
        out = torch.empty((world * cap, width), dtype=local_record.dtype, device=local_record.device)
        dist.all_gather_into_tensor(out, local_record.contiguous(), group=group)
        return out
    padded = torch.zeros((cap, width), dtype=local_record.dtype, device=local_record.device)
    padded[:local_record.shape[0]] = local_record
    out = torch.empty((world * cap, width), dtype=local_record.dtype, device=local_record.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * cap:r * cap + counts[r]] for r in range(world)], dim=0)


def attach_record_comm(engine, group=None, device=None):
    """Give `engine` (blackbox_mpc_amd.engine.Engine) its own RCCL communicator over the ranks of `group`, for the
    device-side record all-gather (Engine.gather_records_dev / gather_wait; include/bbmpc.h "multi-GPU").

    torch.distributed is only the side channel that ships rank 0's 128-byte unique id; the per-control-step
    collective is then enqueued by the engine itself on its communication stream (c10d spends ~30 us of host time
    per asynchronous collective, most of a ~50 us control step -- tools/gather_overhead.py).  `device`: where the
    broadcast tensor lives (a torch.device for the nccl backend, None = CPU for gloo)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    raw = engine.comm_unique_id() if rank == 0 else bytes(128)
    buf = torch.frombuffer(bytearray(raw), dtype=torch.uint8).clone()
    if device is not None:
        buf = buf.to(device)
    src = dist.get_global_rank(group, 0) if group is not None else 0
    dist.broadcast(buf, src=src, group=group)
    engine.comm_init(bytes(buf.cpu().numpy().tobytes()), world, rank)
    return world, rank


_local_keys = iter(range(1, 1 << 62))


def attach_local_comm(engines):
    """One in-process communicator over `engines` (handles of THIS process on one device), engine i = rank i:
    bbmpc_comm_init_local.  What RCCL does between processes on different GPUs, between handles that share a GPU: the
    rank > 0 / nranks > 1 paths of the engine on a one-GPU box (tests/test_gpu_local_ranks.py).  Every collective holds a host
    rendezvous of the ranks: drive each engine from its own thread (ctypes releases the GIL inside the library)."""
    import os
    key = (os.getpid() << 20) ^ next(_local_keys)
    for r, e in enumerate(engines):
        e.comm_init_local(key, len(engines), r)
    return key


class ShardedMPCPolicy:
    """MPCPolicy over all agents with the agents sharded across the ranks of a process group.

    policy_factory(agent_offset, num_agents_local, num_agents_global) must build the rank-local policy
    (an MPCPolicy whose optimizer was created with those three values).  `act` takes the GLOBAL observation
    batch [A_global, S] on every rank and returns the global (action, next_observation, reward) arrays."""

    def __init__(self, policy_factory, num_agents_global, group=None, device=None):
        import torch.distributed as dist
        self._dist = dist
        self._group = group
        self._world = dist.get_world_size(group)
        self._rank = dist.get_rank(group)
        self._A = int(num_agents_global)
        # every rank needs at least one agent: a rank without agents has no record layout to split the gathered rows
        # with, and finding that out inside act() -- after it joined the collectives -- would leave the ranks out of step
        if self._world > self._A:
            raise ValueError("ShardedMPCPolicy: world_size %d > num_agents_global %d (give every rank >= 1 agent; "
                             "single-agent problems run as independent replicas)" % (self._world, self._A))
        self._offset, self._count = agent_shard(self._A, self._world, self._rank)
        self._device = device
        self._policy = policy_factory(self._offset, self._count, self._A) if self._count > 0 else None

    @property
    def local_agents(self):
        return self._offset, self._count

    def reset(self):
        if self._policy is not None:
            self._policy.reset()

    def act(self, observations, t, exploration_noise=False):
        import torch
        obs = np.asarray(observations, dtype=np.float32)
        if obs.ndim != 2 or obs.shape[0] != self._A:
            raise ValueError("observations must be [num_agents_global, dim_S]")
        if self._policy is not None:
            a, n, r = self._policy.act(obs[self._offset:self._offset + self._count], t, exploration_noise)
            rec = np.concatenate([a, n, np.asarray(r, np.float32).reshape(-1, 1)], axis=1).astype(np.float32)
            self._widths = (a.shape[1], n.shape[1])
        else:
            rec = None
        # ranks without agents still take part in the collective: learn the record width from rank 0
        width = torch.tensor([rec.shape[1] if rec is not None else 0], dtype=torch.int64)
        if self._device is not None:
            width = width.to(self._device)
        self._dist.all_reduce(width, op=self._dist.ReduceOp.MAX, group=self._group)
        w = int(width.item())
        if rec is None:
            rec = np.zeros((0, w), np.float32)
        local = torch.from_numpy(rec)
        if self._device is not None:
            local = local.to(self._device)
        full = gather_records(local, self._A, self._group).cpu().numpy()
        if not hasattr(self, "_widths"):
            raise RuntimeError("a rank without agents cannot split the record; give every rank >= 1 agent")
        u, s = self._widths
        return full[:, :u], full[:, u:u + s], full[:, u + s]
