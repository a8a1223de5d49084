"""Row-block / column-block sharding of a block's linears over several ranks (slim_amd/sharded.py) against the
single-GPU path on identical inputs.  The reference has no distributed code: the single-GPU path, itself checked against
the oracle elsewhere, is the yardstick.

What must hold (and is asserted here):
  * 2:4 masks and first-pass codes are those of the single-GPU run bit for bit: masks of blocks are blocks of the mask,
    the summed integer histograms of the blocks are the histogram of the whole, hence the same scale and codes;
  * the factors come from a Gram matrix summed from per-rank fp32 partial sums (instead of one pass) and rounded to bf16,
    so they are not bit-identical.  Tolerance: the error of the fit ||E - (L R)^T / r diag(s)||_F (E = the saliency-weighted
    error matrix of the first pass, identical in every run) of the sharded run within 1e-3 (relative) of the single-GPU
    run's, which is the budget of the fit itself (measured: 1e-6).  The residual weight is then quantised with a scale
    found by the reference's two-stage search (11 coarse candidates, then the bins around the best of them,
    slim/quantization/quantization.py:34-62), which is discontinuous in its input: a residual that differs in a few
    roundings can tip the coarse stage to the neighbouring slot and the scale by several per cent (measured: 135.2 vs
    130.2 on one layer, with the total error 2.3 % higher).  So the second scaling factor is compared to 6 % and the total
    saliency-weighted error ||(W - Wc - (L R)^T / r) diag(s)||_F (dominated by the 4-bit rounding of the residual) to 4 %;
  * one rank running the sharded code alone (no process group) and two ranks both meet that tolerance, and both ranks
    end with identical tensors.
Two ranks share the one GPU of the test box through a gloo group (RCCL refuses two ranks on one device), so the
collectives are staged through the host; which tensors are exchanged is the same."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"

WORKER = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
quant, shard, out_path = sys.argv[2], sys.argv[3] == "1", sys.argv[4]
world = int(os.environ.get("WORLD_SIZE", "1"))
if world > 1:
    dist.init_process_group("gloo")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
from slim_amd import ops, sharded
from slim_amd.prune import compress_linears, block_stats
from slim_amd.layerwrapper import WrappedGPT
from slim_amd.lora import lora_rank
from slim_amd.quantization.quantization import Quantizer

# (name, N, K, outlier channels, their factor): row-sharded, column-sharded, square; with and without deflated columns
SHAPES = [("tall", 1024, 512, 0, 1.0), ("wide", 512, 1024, 0, 1.0), ("square", 512, 512, 0, 1.0),
          ("tall_outl", 1024, 512, 3, 40.0), ("wide_outl", 512, 1024, 3, 40.0), ("wide_huge", 512, 1536, 2, 3000.0)]
items, W0, S = [], {}, {}
for i, (name, n, k, n_out, factor) in enumerate(SHAPES):
    g = torch.Generator(device=dev).manual_seed(10 + i)
    lin = torch.nn.Linear(k, n, bias=False, device=dev, dtype=torch.bfloat16)
    lin.weight.requires_grad_(False)
    lin.weight.data = (torch.randn(n, k, generator=g, device=dev) * 0.02).to(torch.bfloat16)
    W0[name] = lin.weight.data.clone()
    scale = torch.exp(torch.randn(k, generator=g, device=dev) * 0.5)
    if n_out:
        scale[torch.randperm(k, generator=g, device=dev)[:n_out]] *= factor
    w = WrappedGPT(lin)
    w.scaler_row = ops.shift_zeros((scale * scale * 64.0).float())
    S[name] = torch.sqrt(w.scaler_row)
    items.append((name, lin, w))
block_stats([w for _, _, w in items], lambda w: min(lora_rank((w.rows, w.columns), 0.1, 128), w.rows, w.columns))
q = Quantizer("weight", num_bits=4, slim_quant=quant == "slim")
if shard:
    assert sharded.supported(items, world, 2, 4, q, 0.1, True, False, True, True, 128)
compress_linears(items, dev, 2, 4, sparsity_ratio=0.5, quantizer=q, lora_rank=0.1, slim_lora=True, quantize_lora=True,
                 lora_tile_size=128, shard=shard)
out = {}
for name, lin, w in items:
    # the error matrix of the first pass (identical in every run) and what the adapters leave of it
    mask = ops.mask_nm(W0[name], S[name], 2, 4)
    hist = ops.hist_abs(W0[name])
    sf1 = (ops.quant_search(hist, max(512, min(W0[name].numel() // 1000, 20000)), 4)[2] if quant == "slim"
           else ops.absmax_scale(hist, 4)[1]).reshape(1)
    E32 = ops.error_matrix(W0[name], mask, S[name], sf1, 4, want_f32=True)[0]
    approx = (lin.lora_left.data.float() @ lin.lora_right.data.float()).t() / int(lin.lora_rank) * S[name].reshape(1, -1)
    out[name] = dict(fit_err=float(torch.norm(E32 - approx)), E_norm=float(torch.norm(E32)), weight=lin.weight.data.cpu(), L=lin.lora_left.data.cpu(), R=lin.lora_right.data.cpu(),
                     sf=lin.scaling_factor.float().cpu(), sf_dtype=str(lin.scaling_factor.dtype), rank=int(lin.lora_rank),
                     hooks=len(lin._forward_hooks), W0=W0[name].cpu(), s=S[name].cpu(),
                     mask=ops.mask_nm(W0[name], S[name], 2, 4).cpu(),
                     t=0 if w.outliers is None else int(w.outliers[1]), info=lin.lora_fit_info.cpu())
torch.save(out, out_path + "." + os.environ.get("RANK", "0"))
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
'''


def _run(tmp_path, quant, shard, world, port=0):
    worker = tmp_path / "worker.py"
    worker.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = str(tmp_path / f"{quant}_{int(shard)}_{world}.pt")
    args = [str(worker), ROOT, quant, str(int(shard)), out]
    if world == 1:
        cmd = [sys.executable] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
               "127.0.0.1", "--master-port", str(port)] + args
    subprocess.run(cmd, check=True, env=env, timeout=300)
    return [torch.load(out + f".{r}") for r in range(world)]


def _weighted_error(d):
    recon = d["weight"].float() + (d["L"].float() @ d["R"].float()).t() / d["rank"]
    return float(torch.norm((d["W0"].float() - recon) * d["s"].reshape(1, -1)))


@pytest.mark.parametrize("quant,port", [("slim", 29621), ("absmax", 29623)])
def test_sharded_matches_single_gpu(tmp_path, quant, port):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    single = _run(tmp_path, quant, False, 1)[0]
    alone = _run(tmp_path, quant, True, 1)[0]
    two = _run(tmp_path, quant, True, 2, port)
    assert any(d["t"] > 0 for d in single.values()) and any(d["t"] == 0 for d in single.values())
    for name, a in single.items():
        for label, b in (("one rank", alone[name]), ("two ranks, rank 0", two[0][name]), ("two ranks, rank 1", two[1][name])):
            assert b["hooks"] == 1 and b["rank"] == a["rank"] and b["L"].shape == a["L"].shape and b["R"].shape == a["R"].shape
            assert b["sf_dtype"] == a["sf_dtype"] == ("torch.float32" if quant == "slim" else "torch.bfloat16")
            # (the last three words are the fit's reports -- nothing to repeat; measured spread; where the cut sits -- not errors)
            assert int(b["info"][:-3].abs().sum()) == 0 and int(b["info"][-1]) == 0, (name, label, b["info"])
            assert bool((b["weight"][b["mask"]] == 0).all()) and bool((a["weight"][a["mask"]] == 0).all())
            assert torch.equal(a["mask"], b["mask"])
            # kept entries: the same 4-bit level except where the slightly different factors tip a rounding
            differ = (a["weight"] != b["weight"]).float().mean().item()
            ea, eb = _weighted_error(a), _weighted_error(b)
            print(f"{quant} {name} [{label}]: fit error single {a['fit_err']:.5f} sharded {b['fit_err']:.5f} "
                  f"({b['fit_err'] / a['fit_err'] - 1:+.2e}; ||E|| {a['E_norm']:.3f}), total error {ea:.5f} / {eb:.5f} "
                  f"({eb / ea - 1:+.2e}), sf {float(a['sf']):.4f} / {float(b['sf']):.4f}, deflated {a['t']}, "
                  f"entries differing {differ:.3f}")
            assert abs(b["fit_err"] / a["fit_err"] - 1) <= 1e-3, (name, label, a["fit_err"], b["fit_err"])
            assert abs(eb / ea - 1) <= 4e-2, (name, label, ea, eb)
            assert abs(float(b["sf"]) / float(a["sf"]) - 1) <= 6e-2, (name, label)
        for key in ("weight", "L", "R", "sf"):       # both ranks hold the same compressed linear
            assert torch.equal(two[0][name][key], two[1][name][key]), (name, key)


def test_block_stages_are_bit_exact():
    """Masks of row / column blocks are the blocks of the mask, the block histograms add up to the whole histogram (so the
    scale search and the first-pass codes are those of the single-GPU run), for 2 to 8 ranks and ragged block sizes."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from slim_amd import ops
    from slim_amd.sharded import shard_bounds
    g = torch.Generator(device=DEV).manual_seed(3)
    for N, K in ((1408, 512), (512, 1408)):                 # 11 aligned blocks on the long side: ragged for every world
        W = (torch.randn(N, K, generator=g, device=DEV) * 0.02).to(torch.bfloat16)
        s = torch.exp(torch.randn(K, generator=g, device=DEV))
        mask = ops.mask_nm(W, s, 2, 4)
        hist = ops.hist_abs(W)
        sf = ops.quant_search(hist, 720, 4)[2].reshape(1)
        codes = ops.quantize_pt(W, sf, 4)
        _, E = ops.error_matrix(W, mask, s, sf, 4)
        for world in (2, 3, 8):
            acc = torch.zeros_like(hist)
            bounds = shard_bounds(max(N, K), world)
            assert bounds[0][0] == 0 and bounds[-1][1] == max(N, K) and all(a[1] == b[0] for a, b in zip(bounds, bounds[1:]))
            for lo, hi in bounds:
                if K <= N:
                    Ws, ss, sl = W[lo:hi].contiguous(), s, (slice(lo, hi), slice(None))
                else:
                    Ws, ss, sl = W[:, lo:hi].contiguous(), s[lo:hi].contiguous(), (slice(None), slice(lo, hi))
                ms = ops.mask_nm(Ws, ss, 2, 4)
                assert torch.equal(ms, mask[sl])
                ops.hist_abs(Ws, acc)
                assert torch.equal(ops.quantize_pt(Ws, sf, 4), codes[sl])
                assert torch.equal(ops.error_matrix(Ws, ms, ss, sf, 4)[1], E[sl])
            assert torch.equal(acc, hist)


def test_sharded_code_path_at_13b_shape_within_1e3_of_truncated_svd():
    """The pieces of the sharded fit (Gram contribution, slim_gramfit_iterate / project / factors, batched eigen-solver) on a
    LLaMA-2-13B-shaped linear (5120 x 13824, rank 512, column blocks): the adapters' fit error against the exact truncated
    SVD of the error matrix (slim/lora.py:172-185), budget 1e-3 as for the single-GPU path.  One rank, so no exchange; the
    two-rank run of the same code is the test above."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from slim_amd import ops, sharded
    from slim_amd.prune import block_stats
    from slim_amd.layerwrapper import WrappedGPT
    from slim_amd.quantization.quantization import Quantizer
    N, K, rank = 5120, 13824, 512
    g = torch.Generator(device=DEV).manual_seed(5)
    lin = torch.nn.Linear(K, N, bias=False, device=DEV, dtype=torch.bfloat16)
    lin.weight.requires_grad_(False)
    lin.weight.data = (torch.randn(N, K, generator=g, device=DEV) * 0.02).to(torch.bfloat16)
    W0 = lin.weight.data.clone()
    scale = torch.exp(torch.randn(K, generator=g, device=DEV) * 0.5)
    scale[torch.randperm(K, generator=g, device=DEV)[:4]] *= 30.0
    w = WrappedGPT(lin)
    w.scaler_row = ops.shift_zeros((scale * scale * 2048.0).float())
    s = torch.sqrt(w.scaler_row)
    block_stats([w], lambda w: rank)
    q = Quantizer("weight", num_bits=4, slim_quant=True)
    items = [("down_proj", lin, w)]
    assert sharded.supported(items, 8, 2, 4, q, 0.1, True, False, True, True, 256)      # 108 aligned column blocks for 8 ranks
    status = sharded.compress_linears_sharded(items, torch.device(DEV), 2, 4, q, 0.1, True, 256)
    assert int(status[:, :2].abs().sum()) == 0              # (columns 2, 3: the measured spread and where the cut sits)
    assert int(lin.lora_rank) == rank and lin.lora_left.shape == (K, rank) and lin.lora_right.shape == (rank, N)
    mask = ops.mask_nm(W0, s, 2, 4)
    sf1 = ops.quant_search(ops.hist_abs(W0), 20000, 4)[2].reshape(1)
    E32 = ops.error_matrix(W0, mask, s, sf1, 4, want_f32=True)[0]
    approx = (lin.lora_left.data.float() @ lin.lora_right.data.float()).t() / rank * s.reshape(1, -1)
    err = float(torch.norm(E32 - approx))
    sv = torch.linalg.svdvals(E32)
    opt = float(torch.sqrt((sv[rank:].double() ** 2).sum()))
    print(f"5120x13824 sharded code path: fit error {err:.3f}, truncated SVD {opt:.3f}, excess {err / opt - 1:+.2e}, "
          f"deflated {0 if w.outliers is None else w.outliers[1]}")
    assert err / opt - 1 <= 1e-3
    assert bool((lin.weight.data[mask] == 0).all())
