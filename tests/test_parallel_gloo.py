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
    # the same exchange with every owner's tensors in ONE broadcast (deferred form + broadcast_coalesced): mixed dtypes,
    # byte counts that are no multiple of the buffer's alignment (the 4-byte scales), identical results
    from slim_amd.parallel import broadcast_coalesced
    outgoing, completions, lins2 = {}, [], []
    for i, (n, k, r) in enumerate(shapes):
        lin = torch.nn.Linear(k, n, bias=False).to(torch.bfloat16)
        if owner[i] == rank:
            for name in ("lora_left", "lora_right", "scaling_factor"):
                setattr(lin, name, getattr(lins[i], name))
            lin.weight.data = lins[i].weight.data.clone()
        if owner[i] == rank:
            lin.lora_left_mask = torch.arange(k * r).reshape(k, r) % (i + 2) == 0
        completions.append(exchange_linear(lin, owner[i], rank, (n, k, r), "cpu", has_left_mask=True,
                                           defer=outgoing.setdefault(owner[i], [])))
        lins2.append(lin)
    for src in sorted(outgoing):
        broadcast_coalesced(outgoing[src], src)
    for complete in completions:
        complete()
    for i in range(len(shapes)):
        for name in ("lora_left", "lora_right", "scaling_factor"):
            ok &= torch.equal(getattr(lins2[i], name).data, getattr(lins[i], name).data)
        ok &= torch.equal(lins2[i].weight.data, lins[i].weight.data)
        n, k, r = shapes[i]
        ok &= torch.equal(lins2[i].lora_left_mask, torch.arange(k * r).reshape(k, r) % (i + 2) == 0)
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


# ---- token-sharded calibration Gram matrix -----------------------------------------------------------------------------
def _gram_inputs():
    g = torch.Generator().manual_seed(77)
    scale = torch.exp(torch.randn(48, generator=g) * 0.5)
    return (torch.randn(6, 40, 48, generator=g) * scale).to(torch.bfloat16)          # 6 samples of 40 tokens, K = 48


def _gram_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cpu_standin_ops as standin
    import slim_amd.layerwrapper as LW
    from slim_amd import parallel
    LW.ops = standin                                         # (no GPU here: the kernels' CPU stand-ins, same arithmetic)
    X = _gram_inputs()
    # an uneven split of the samples: rank 0 takes four (in two forward calls), rank 1 two (one at a time)
    mine = [X[0:3], X[3:4]] if rank == 0 else [X[4:5], X[5:6]]
    w = LW.WrappedGPT(torch.nn.Linear(48, 16, bias=False).to(torch.bfloat16), collect_gram=True)
    for chunk in mine:
        w.add_batch(chunk, None, per_sample=True)
    local_n = w.gram_nsamples
    w.reduce_gram(parallel.all_reduce)
    q.put((rank, local_n, w.gram_nsamples, w.H.clone(), w.scaler_row.clone(), w.nsamples))
    dist.destroy_process_group()


def test_token_sharded_gram_matches_one_rank():
    """WrappedGPT(collect_gram=True) on two ranks, each folding its share of the samples, then reduce_gram (one all-reduce
    SUM of the count-weighted matrices): equal to one rank folding all samples (slim/sparsegpt.py:28-39) within 1e-5, and to
    the oracle's recurrence over the samples one at a time."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import cpu_standin_ops as standin
    import slim_amd.layerwrapper as LW
    from oracle import slim_oracle as O
    X = _gram_inputs()
    real = LW.ops
    LW.ops = standin
    try:
        one = LW.WrappedGPT(torch.nn.Linear(48, 16, bias=False).to(torch.bfloat16), collect_gram=True)
        one.add_batch(X, None, per_sample=True)
        assert one.reduce_gram(None) is one.H and one.gram_nsamples == 6
        with pytest.raises(ValueError):
            LW.WrappedGPT(torch.nn.Linear(48, 16, bias=False).to(torch.bfloat16)).reduce_gram(None)
    finally:
        LW.ops = real
    Href, n = torch.zeros(48, 48), 0
    for j in range(6):
        Href, n = O.add_batch_gram(Href, n, X[j])
    scale = float(Href.abs().max())
    torch.testing.assert_close(one.H, Href, rtol=1e-5, atol=1e-5 * scale)
    torch.testing.assert_close(torch.diag(one.H) / 2, one.scaler_row, rtol=1e-5, atol=0)      # diag(H) / 2 == scaler_row
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gram_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [4, 2] and all(r[2] == 6 for r in res)
    assert torch.equal(res[0][3], res[1][3])                                       # the same matrix on both ranks
    torch.testing.assert_close(res[0][3], one.H, rtol=1e-5, atol=1e-5 * scale)
    # (scaler_row stays the rank's own: the sharded driver gathers the per-sample sums instead, see gather_sample_stats)
    assert res[0][5] == 4 and res[1][5] == 2
