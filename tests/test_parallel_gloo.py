"""The N > 1 path on CPU: partitioning is deterministic and balanced, and a world-size-2 gloo run leaves
every rank with every compressed linear after the owner broadcasts."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from slim_amd.parallel import partition_by_cost, linear_cost, exchange_linear, gather_sample_stats

BLOCK = [(4096, 4096)] * 4 + [(11008, 4096)] * 2 + [(4096, 11008)]


def test_partition_balanced_and_deterministic():
    costs = [linear_cost(n, k) for n, k in BLOCK]
    for world in (1, 2, 4, 8):
        owner = partition_by_cost(costs, world)
        assert owner == partition_by_cost(costs, world) and set(owner) <= set(range(world))
        loads = [sum(c for c, o in zip(costs, owner) if o == r) for r in range(world)]
        # LPT bound: no rank carries more than the mean load plus the largest item
        assert max(loads) <= sum(costs) / world + max(costs)
        if world >= 7:
            assert sorted(owner) == sorted(set(owner))       # one linear per rank


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shapes = [(64, 32, 16), (32, 64, 16), (48, 32, 8)]
    owner = partition_by_cost([n * k for n, k, _ in shapes], world)
    lins = []
    for i, (n, k, r) in enumerate(shapes):
        lin = torch.nn.Linear(k, n, bias=False).to(torch.bfloat16)
        if owner[i] == rank:      # stand-in for the compression result of the owner
            g = torch.Generator().manual_seed(1000 + i)
            lin.weight.data = torch.randn(n, k, generator=g).to(torch.bfloat16)
            lin.lora_left = torch.nn.Parameter(torch.randn(k, r, generator=g).to(torch.bfloat16), requires_grad=False)
            lin.lora_right = torch.nn.Parameter(torch.randn(r, n, generator=g).to(torch.bfloat16), requires_grad=False)
            lin.scaling_factor = torch.full((1, 1), 100.0 + i)
        exchange_linear(lin, owner[i], rank, (n, k, r), "cpu")
        lins.append(lin)
    ok = True
    for i, (n, k, r) in enumerate(shapes):
        g = torch.Generator().manual_seed(1000 + i)
        ok &= torch.equal(lins[i].weight.data, torch.randn(n, k, generator=g).to(torch.bfloat16))
        ok &= torch.equal(lins[i].lora_left.data, torch.randn(k, r, generator=g).to(torch.bfloat16))
        ok &= torch.equal(lins[i].lora_right.data, torch.randn(r, n, generator=g).to(torch.bfloat16))
        ok &= float(lins[i].scaling_factor) == 100.0 + i
    # sample-ordered statistics gather
    local = torch.full((2, 8), float(rank)) + torch.arange(2).reshape(2, 1)
    allst = gather_sample_stats(local)
    ok &= allst.shape == (2 * world, 8) and torch.equal(allst[:, 0], torch.tensor([0.0, 1.0, 1.0, 2.0]))
    q.put((rank, bool(ok), owner))
    dist.destroy_process_group()


def test_exchange_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert res[0][2] == res[1][2] and set(res[0][2]) == {0, 1}
