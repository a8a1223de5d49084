"""The row-block sharded path (slim_amd/sharded.py) on CPU: block bounds, the staged collectives, and the whole
orchestration over a world-size-2 gloo group with the kernels replaced by CPU stand-ins (tests/cpu_standin_ops.py), against
the same code on one rank and against the oracle's add_lora (exact SVD, slim/lora.py:105-213).

Tolerances, stated: masks bit-exact.  The first scaling factor bit-exact (integer histograms add up).  Two ranks vs one:
the Gram matrix and the Rayleigh-Ritz matrix are sums of two fp32 partial sums instead of one pass, which moves bf16
roundings of the basis: the error of the fit ||E - U L^T||_F agrees to 1e-4 relative.  Against the exact SVD: 1e-3."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from slim_amd.sharded import shard_bounds

SHAPES = [("tall", 384, 128, 0), ("wide", 128, 384, 0), ("square", 256, 256, 0), ("tall_outl", 384, 128, 2),
          ("wide_outl", 128, 512, 2)]


def test_shard_bounds():
    for n in (128, 1408, 4096, 11008, 13824, 28672):
        for world in (1, 2, 3, 4, 8):
            if n // 128 < world:
                continue
            b = shard_bounds(n, world)
            assert len(b) == world and b[0][0] == 0 and b[-1][1] == n
            assert all(x[1] == y[0] for x, y in zip(b, b[1:]))
            sizes = [hi - lo for lo, hi in b]
            assert all(s % 128 == 0 and s > 0 for s in sizes) and max(sizes) - min(sizes) <= 128
    with pytest.raises(ValueError):
        shard_bounds(1000, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_items():
    from slim_amd.layerwrapper import WrappedGPT
    from slim_amd.lora import lora_rank, _outliers_from_host, MAX_OUTLIER_COLS
    items, W0 = [], {}
    for i, (name, n, k, n_out) in enumerate(SHAPES):
        g = torch.Generator().manual_seed(50 + i)
        lin = torch.nn.Linear(k, n, bias=False).to(torch.bfloat16)
        lin.weight.requires_grad_(False)
        lin.weight.data = (torch.randn(n, k, generator=g) * 0.02).to(torch.bfloat16)
        W0[name] = lin.weight.data.clone()
        scale = torch.exp(torch.randn(k, generator=g) * 0.5)
        if n_out:
            scale[torch.randperm(k, generator=g)[:n_out]] *= 100.0
        w = WrappedGPT(lin)
        w.scaler_row = (scale * scale * 64.0).float()
        # what prune.block_stats derives on the GPU
        s = torch.sqrt(w.scaler_row)
        top = torch.topk(s, min(MAX_OUTLIER_COLS + 1, k))
        w.has_zero = False
        w.outliers, w.scale_spread = _outliers_from_host(top.values.tolist(), float(s.median()), top.indices,
                                                         min(lora_rank((n, k), 0.25, None), n, k))
        items.append((name, lin, w))
    return items, W0


def _compress(items, world_expected):
    import cpu_standin_ops as standin
    from slim_amd import sharded, parallel, lora
    from slim_amd.quantization.quantization import Quantizer
    sharded.ops = standin
    lora.ops = standin          # (the Rayleigh-Ritz stage is lora.rayleigh_ritz)
    q = Quantizer("weight", num_bits=4, slim_quant=True)
    assert parallel.world_info()[1] == world_expected
    assert sharded.supported(items, world_expected, 2, 4, q, 0.25, True, False, True, True, None)
    status = sharded.compress_linears_sharded(items, torch.device("cpu"), 2, 4, q, 0.25, True, None)
    assert int(status[:, :2].abs().sum()) == 0
    out = {}
    for name, lin, w in items:
        assert len(lin._forward_hooks) == 1 and int(lin.lora_rank) == lin.lora_left.shape[1] == lin.lora_right.shape[0]
        out[name] = dict(weight=lin.weight.data.clone(), L=lin.lora_left.data.clone(), R=lin.lora_right.data.clone(),
                         sf=lin.scaling_factor.clone(), rank=int(lin.lora_rank),
                         t=0 if w.outliers is None else int(w.outliers[1]))
    return out


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    items, _ = _make_items()
    out = _compress(items, world)
    # the staged helpers on plain tensors
    from slim_amd import parallel
    from slim_amd.sharded import _gather_blocks
    t = torch.full((3,), float(rank + 1))
    ok = torch.equal(parallel.all_reduce(t.clone()), torch.full((3,), 3.0))
    ok &= torch.equal(parallel.all_reduce(t.clone(), op="max"), torch.full((3,), 2.0))
    red = parallel.reduce(t.clone(), 1)
    ok &= rank != 1 or torch.equal(red, torch.full((3,), 3.0))
    bounds = shard_bounds(384, 2)                      # 256 + 128: ragged
    whole = torch.arange(384 * 4, dtype=torch.float32).reshape(384, 4).to(torch.bfloat16)
    lo, hi = bounds[rank]
    ok &= torch.equal(_gather_blocks(whole[lo:hi].contiguous(), bounds, 0), whole)
    ok &= torch.equal(_gather_blocks(whole.t()[:, lo:hi].contiguous(), bounds, 1), whole.t().contiguous())
    # tensors sent through a multiprocessing queue live in shared memory that dies with this process: send arrays
    q.put((rank, bool(ok), {n: {k: (v.float().numpy() if isinstance(v, torch.Tensor) else v) for k, v in d.items()}
                            for n, d in out.items()}))
    dist.barrier()
    dist.destroy_process_group()


def _fit_error(W0, s, d, sf1):
    """||E - U L^T diag(s)||_F for the adapters in d, E as slim/lora.py:146-156."""
    from oracle import slim_oracle as O
    mask = O.prune_nm(W0.abs() * s.reshape(1, -1), 2, 4)
    codes = torch.clamp(torch.round((W0 * sf1).float()), -8, 7)
    new_w = codes / sf1 * s.reshape(1, -1)
    new_w[mask] = 0
    E = (W0 * s.reshape(1, -1) - new_w).float()
    approx = (d["L"].float() @ d["R"].float()).t() / d["rank"] * s.reshape(1, -1)
    return float(torch.norm(E - approx)), mask, E


def test_sharded_world_2_gloo_equals_one_rank_and_oracle():
    from oracle import slim_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    two0, two1 = ({n: {k: (torch.from_numpy(v).to(torch.float32 if k == "sf" else torch.bfloat16) if k in ("weight", "L", "R", "sf")
                           else v) for k, v in d.items()} for n, d in r[2].items()} for r in res)
    items, W0 = _make_items()
    scales = {name: torch.sqrt(w.scaler_row) for name, _, w in items}
    one = _compress(items, 1)
    assert any(d["t"] > 0 for d in one.values()) and any(d["t"] == 0 for d in one.values())
    for name in one:
        for key in ("weight", "L", "R", "sf"):           # both ranks hold the same compressed linear
            assert torch.equal(two0[name][key], two1[name][key]), (name, key)
        W, s = W0[name], scales[name]
        sf1 = 7.0 / O.find_optimal_quantization_cap(W, 4, O.num_bins_for(W.numel()))
        e1, mask, E = _fit_error(W, s, one[name], sf1)
        e2, _, _ = _fit_error(W, s, two0[name], sf1)
        # the oracle: exact truncated SVD of E (slim/lora.py:172-185)
        sv = torch.linalg.svdvals(E.double())
        opt = float(torch.sqrt((sv[one[name]["rank"]:] ** 2).sum()))
        for d in (one[name], two0[name]):
            assert bool((d["weight"][mask] == 0).all())
            kept_zero = ((d["weight"] == 0) & ~mask).float().mean().item()
            assert kept_zero < 0.2
        print(f"{name}: fit error one rank {e1:.5f}, two ranks {e2:.5f} ({e2 / e1 - 1:+.2e}), truncated SVD {opt:.5f} "
              f"({e1 / opt - 1:+.2e}), deflated {one[name]['t']}")
        assert abs(e2 / e1 - 1) <= 1e-4, (name, e1, e2)
        assert e1 / opt - 1 <= 1e-3 and e2 / opt - 1 <= 1e-3, (name, e1, e2, opt)
        assert abs(float(two0[name]["sf"]) / float(one[name]["sf"]) - 1) <= 2e-2
