"""CPU, world_size 2, gloo: the N>1 path of the APG update -- per-device nan_to_num + clip, ONE mean
all-reduce of the flat gradient bucket, identical Adam steps on every rank (apg.py:233-258)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_grads(rank, policy):
    g = torch.Generator().manual_seed(100 + rank)
    obs = torch.randn(4, 20, generator=g)
    out = policy(obs)
    return (out ** 2).sum() * (1.0 + 50.0 * rank)     # rank 1 exceeds the clip threshold, rank 0 does not


def _worker(rank, world, port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unidom_amd.algorithms.apg.core import GradSync, Policy
    torch.manual_seed(0)
    policy = Policy(20, 3, hidden=(16, 8), seed=1)
    sync = GradSync(policy, learning_rate=1e-2, max_gradient_norm=0.3)
    for _ in range(2):
        sync.zero_grad()
        _make_grads(rank, policy).backward()
        if rank == 1:
            sync.flat_grad[3] = float("nan")            # nan_to_num happens per device before the clip
        raw = sync.step()
    out_q.put((rank, torch.cat([p.detach().reshape(-1) for p in policy.parameters()]).numpy(), float(raw)))
    dist.barrier()
    dist.destroy_process_group()


def _reference(world):
    """Single-process restatement of the same two updates."""
    from unidom_amd.algorithms.apg.core import Policy
    policies = [Policy(20, 3, hidden=(16, 8), seed=1) for _ in range(world)]
    opts = [torch.optim.Adam(p.parameters(), lr=1e-2, betas=(0.9, 0.999), eps=1e-8) for p in policies]
    for _ in range(2):
        flats = []
        for r, pol in enumerate(policies):
            for p in pol.parameters():
                p.grad = None
            _make_grads(r, pol).backward()
            g = torch.cat([p.grad.reshape(-1) for p in pol.parameters()])
            if r == 1:
                g[3] = float("nan")
            g = torch.nan_to_num(g)
            n = torch.linalg.vector_norm(g)
            g = g if n < 0.3 else g / n * 0.3
            flats.append(g)
        mean = sum(flats) / world
        for pol, opt in zip(policies, opts):
            off = 0
            for p in pol.parameters():
                p.grad = mean[off:off + p.numel()].view_as(p).clone()
                off += p.numel()
            opt.step()
    return torch.cat([p.detach().reshape(-1) for p in policies[0].parameters()]).numpy()


@pytest.mark.timeout(120)
def test_two_rank_gradient_mean_and_replicated_update():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict()
    for _ in range(world):
        r, params, raw = q.get(timeout=100)
        res[r] = (params, raw)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    np.testing.assert_array_equal(res[0][0], res[1][0])                  # replicas stay identical
    np.testing.assert_allclose(res[0][0], _reference(world), rtol=1e-6, atol=1e-7)
    assert res[1][1] > 0.3 > 0                                           # rank 1 really was clipped


def test_env_sharding_rule():
    from unidom_amd.algorithms.apg.core import shard_envs
    assert shard_envs(256, 8) == 32 and shard_envs(4, 1) == 4
    with pytest.raises(ValueError):
        shard_envs(6, 4)
