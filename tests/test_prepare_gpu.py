"""The first phase of a linear as one library call (slim_lora_prepare) against the same steps as separate calls
(lora.add_lora_prepare + add_lora_deflate): mask, scale, error matrix and its transpose bit for bit; the split-off of the
outlier columns as the same decomposition E = Q_O (Q_O^T E) + E' with an orthonormal Q_O of the same span."""
import types

import pytest
import torch

from oracle import slim_oracle as O
from conftest import synth_weight, synth_scaler

pytestmark = pytest.mark.gpu


def _wrapped(sr, rank):
    from slim_amd.prune import block_stats
    w = types.SimpleNamespace(scaler_row=sr.cuda(), rows=0, columns=0)
    block_stats([w], lambda _: rank)
    return w


@pytest.mark.parametrize("shape,slim_quant,outliers", [((512, 768), True, True), ((768, 512), False, True),
                                                        ((1024, 1024), True, False), ((256, 2048), True, True)])
def test_fused_prepare_equals_separate_calls(shape, slim_quant, outliers):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from slim_amd import ops, lora
    from slim_amd.quantization.quantization import Quantizer
    N, K = shape
    W = synth_weight(N, K, 900 + N % 7, heavy=True).cuda()
    sr = O.shift_zeros(synth_scaler(K, 910 + K % 5))
    if not outliers:
        sr = sr.clamp(max=float(sr.median()) * 4.0)
    lin = torch.nn.Linear(K, N, bias=False, device="cuda", dtype=torch.bfloat16)
    lin.weight.requires_grad_(False)
    lin.weight.data = W
    rank = lora.lora_rank((N, K), 0.1, 256)
    wrapped = _wrapped(sr, min(rank, N, K))
    assert (wrapped.outliers is not None) == outliers
    q = Quantizer("weight", num_bits=4, slim_quant=slim_quant)
    # separate calls
    mask = ops.mask_nm(W, ops.sqrt(wrapped.scaler_row), 2, 4)
    ref = lora.add_lora_prepare(lin, mask, 0.1, True, wrapped, q, lora_tile_size=256, quantize_first=True)
    lora.add_lora_deflate(ref)
    # one call
    got = lora.add_lora_prepare_fused(lin, 2, 4, 0.1, True, wrapped, q, 256)
    torch.cuda.synchronize()
    assert torch.equal(got.W_mask, ref.W_mask)
    assert torch.equal(got.s, ref.s)
    assert torch.equal(got.E, ref.E) and torch.equal(got.Et, ref.Et)
    assert got.rank == ref.rank and got.spread == ref.spread
    assert (got.deflated is None) == (ref.deflated is None) == (not outliers)
    if not outliers:
        return
    E2, E2t, UO, Ct, info = got.deflated
    t = got.outliers[1]
    assert int(info.abs().sum()) == 0
    assert torch.equal(E2t, E2.t().contiguous())
    # orthonormal basis (to bf16 rounding) of the outlier columns, zero padding columns
    G = UO.float().t() @ UO.float()
    eye = torch.zeros_like(G)
    eye[:t, :t] = torch.eye(t, device="cuda")
    assert float((G - eye).abs().max()) < 2e-2
    cols = got.outliers[0][:t].long()
    EO = got.E.float()[:, cols]
    proj = UO.float() @ (UO.float().t() @ EO)
    assert float((proj - EO).norm() / EO.norm()) < 2e-2          # the basis spans the outlier columns
    # the decomposition: E = UO Ct^T + E2 up to the bf16 rounding of its three terms
    rebuilt = UO.float() @ Ct.float().t() + E2.float()
    assert float((rebuilt - got.E.float()).norm() / got.E.float().norm()) < 1e-2
    # the separate-call path runs the same kernels: identical tensors
    for a, b in zip(got.deflated[:4], ref.deflated[:4]):
        assert torch.equal(a, b)


def test_fused_prepare_with_given_mask_and_no_quantiser():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from slim_amd import ops, lora
    N, K = 384, 512
    W = synth_weight(N, K, 931).cuda()
    sr = O.shift_zeros(synth_scaler(K, 932))
    lin = torch.nn.Linear(K, N, bias=False, device="cuda", dtype=torch.bfloat16)
    lin.weight.requires_grad_(False)
    lin.weight.data = W
    wrapped = _wrapped(sr, 38)
    mask = (torch.rand(N, K, device="cuda") < 0.5)
    ref = lora.add_lora_prepare(lin, mask, 0.1, True, wrapped, None)
    got = lora.add_lora_prepare_fused(lin, 2, 4, 0.1, True, wrapped, None, None, W_mask=mask)
    torch.cuda.synchronize()
    assert torch.equal(got.W_mask, mask)
    assert torch.equal(got.E, ref.E) and torch.equal(got.Et, ref.Et)


@pytest.mark.parametrize("N,t,tp", [(768, 1, 8), (1024, 13, 16), (4096, 64, 64), (520, 40, 40)])
def test_outlier_basis_is_orthonormal_and_spans_the_columns(N, t, tp):
    """slim_outlier_basis / slim_outlier_project on their own, up to the 64 columns the driver may split off: an orthonormal
    basis (to bf16 rounding) of the listed columns whatever their scales, zero padding columns, and the projection of every
    column of E on it; a repeated column drops out instead of poisoning the basis."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from slim_amd import ops
    g = torch.Generator(device="cuda").manual_seed(N + t)
    K = 256
    E = torch.randn(N, K, device="cuda", generator=g)
    cols = torch.randperm(K, device="cuda", generator=g)[:tp].to(torch.int32)
    E[:, cols[:t].long()] *= torch.logspace(0, 3, t, device="cuda")            # scales three orders of magnitude apart
    E = E.to(torch.bfloat16)
    Et = E.t().contiguous()
    UO, UOt = ops.outlier_basis(Et, cols, t)
    torch.cuda.synchronize()
    assert torch.equal(UOt, UO.t().contiguous())
    assert float(UO[:, t:].abs().max()) == 0.0 if tp > t else True
    G = UO[:, :t].float().t() @ UO[:, :t].float()
    assert float((G - torch.eye(t, device="cuda")).abs().max()) < 3e-2
    EO = E.float()[:, cols[:t].long()]
    proj = UO.float() @ (UO.float().t() @ EO)
    assert float(((proj - EO).norm(dim=0) / EO.norm(dim=0)).max()) < 3e-2
    Ct = ops.outlier_project(Et, UOt)
    ref = (Et.float() @ UO.float())
    assert float((Ct.float() - ref).abs().max()) <= 2 ** -7 * float(ref.abs().max()) + 1e-6
    if t >= 2:
        # the second listed column equal to the first: its direction is dropped, the others stay orthonormal
        cols2 = cols.clone()
        cols2[1] = cols2[0]
        UO2, _ = ops.outlier_basis(Et, cols2, t)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(UO2.float()).all())
        assert float(UO2[:, 1].abs().max()) == 0.0
        keep = [0] + list(range(2, t))
        G2 = UO2[:, keep].float().t() @ UO2[:, keep].float()
        assert float((G2 - torch.eye(len(keep), device="cuda")).abs().max()) < 3e-2
