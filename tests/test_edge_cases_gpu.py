"""Degenerate and boundary inputs of the path, HIP kernels against the oracle: dead input channels (the naive
branch of add_lora), adapters without a weight quantiser, all-zero and constant weights, one-row / minimum-width
layers, ranks that are not multiples of 8, and the argument errors of the C ABI."""
import pytest
import torch

from oracle import slim_oracle as O
from conftest import synth_weight, synth_scaler

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from slim_amd import ops as _ops
    return _ops


class Stats:
    """Stand-in for the WrappedGPT of a layer: only scaler_row is read by add_lora."""

    def __init__(self, scaler_row):
        self.scaler_row = scaler_row


def _weighted_err(W, out, sr, use_s):
    s = torch.sqrt(sr).reshape(1, -1) if use_s else torch.ones(1, W.shape[1])
    eff = out["weight"].float() + out["lora_right"].float().t() @ out["lora_left"].float().t()
    return torch.norm((W.float() - eff) * s).item()


@pytest.mark.parametrize("zeros,quantize", [(5, True), (0, False), (3, False)])
def test_add_lora_branches_against_oracle(ops, zeros, quantize):
    """zeros > 0: statistics with dead channels select the naive branch (lora.py:139, :158-169); quantize False:
    adapters for a pruned-only layer (quantizer None, lora.py:140-144)."""
    from slim_amd.lora import add_lora
    from slim_amd.quantization.quantization import Quantizer
    N, K = 256, 512
    W = synth_weight(N, K, 7)
    sr = synth_scaler(K, 8, zeros=zeros)
    mask = O.prune_nm(W.abs().float() * torch.sqrt(sr).reshape(1, -1), 2, 4)
    ref = O.add_lora(W, mask, sr, 0.1, num_bits=4, tile=None, slim_lora=True, quantize=quantize)
    lin = torch.nn.Linear(K, N, bias=False).to(torch.bfloat16).to(DEV)
    lin.weight.data = W.to(DEV)
    q = Quantizer("weight", num_bits=4, slim_quant=True) if quantize else None
    add_lora(lin, W_mask=mask.to(DEV), rank_ratio=0.1, slim_lora=True, activations=Stats(sr.to(DEV)), quantizer=q,
             lora_tile_size=None, quantize_first=True)
    r = ref["rank"]
    assert r % 8 != 0 and tuple(lin.lora_left.shape) == (K, r) and tuple(lin.lora_right.shape) == (r, N)
    ours = dict(weight=lin.weight.data.cpu(), lora_left=lin.lora_left.data.cpu(), lora_right=lin.lora_right.data.cpu())
    use_s = zeros == 0
    eo, er = _weighted_err(W, ours, sr, use_s), _weighted_err(W, ref, sr, use_s)
    print(f"zeros={zeros} quantize={quantize}: error ours {eo:.5f} oracle {er:.5f}")
    assert eo <= er * 1.01
    assert torch.equal(ours["weight"][mask], torch.zeros_like(ours["weight"][mask]))
    if not quantize:
        assert not hasattr(lin, "scaling_factor") or lin.scaling_factor is None
    else:
        assert ours["weight"].float().unique().numel() <= 16


def test_zero_and_constant_weights(ops):
    """All-zero weights: every group is a four-way tie, the histogram has one bin; constant weights likewise."""
    s = torch.sqrt(O.shift_zeros(synth_scaler(64, 3))).to(DEV)
    for value in (0.0, 0.25):
        W = torch.full((128, 64), value).to(torch.bfloat16)
        want = O.prune_nm(W.abs().float() * s.cpu().reshape(1, -1), 2, 4)
        got = ops.mask_nm(W.to(DEV), s, 2, 4).cpu()
        assert torch.equal(got, want)
        hist = ops.hist_abs(W.to(DEV)).cpu()
        assert int(hist.sum()) == W.numel() and int((hist != 0).sum()) == 1


def test_minimum_shapes(ops):
    """One output row, eight input channels (the narrowest the 16-byte loads allow)."""
    W = synth_weight(1, 8, 5)
    sr = O.shift_zeros(synth_scaler(8, 6))
    s = torch.sqrt(sr)
    assert torch.equal(ops.mask_nm(W.to(DEV), s.to(DEV), 2, 4).cpu(), O.prune_nm(W.abs().float() * s.reshape(1, -1), 2, 4))
    assert torch.equal(ops.mask_unstructured(W.to(DEV), s.to(DEV), 4).cpu(),
                       O.prune_unstructured(W.abs().float() * s.reshape(1, -1), 0.5))
    x = (torch.randn(3, 8) * 2).to(torch.bfloat16)
    sr0, n = torch.zeros(8), 0
    for j in range(3):
        sr0, n = O.add_batch_scaler_row(sr0, n, x[j:j + 1])
    got = torch.zeros(8, device=DEV)
    ops.scaler_row_accum(x.to(DEV), got, 0, 3, 1)
    torch.testing.assert_close(got.cpu(), sr0, rtol=3e-6, atol=0)


def test_argument_errors(ops):
    W = synth_weight(16, 12, 1).to(DEV)            # 12 input channels: not a multiple of 8
    s = torch.ones(12, device=DEV)
    with pytest.raises(ValueError):
        ops.error_matrix(W, torch.zeros(16, 12, dtype=torch.bool, device=DEV), s, None, 4)
    with pytest.raises((ValueError, NotImplementedError)):
        ops.mask_nm(W[:, :10].contiguous(), s[:10].contiguous(), 2, 4)      # K not a multiple of M = 4
    with pytest.raises(NotImplementedError):
        ops.mask_nm(synth_weight(16, 16, 2).to(DEV), torch.ones(16, device=DEV), 4, 8)   # only M = 4 is built
    with pytest.raises(ValueError):
        ops.lowrank_fit(synth_weight(64, 64, 3).to(DEV), synth_weight(64, 64, 3).to(DEV).t().contiguous(), 60, 8, 2)  # r % 8
    with pytest.raises(ValueError):
        ops.mask_unstructured(synth_weight(16, 16, 2).to(DEV), torch.ones(16, device=DEV), 17)


def test_stream_feeding_modes_agree(ops):
    """compress_linears fed by one host thread per stream, by the calling thread alone, and with the per-job timeline on:
    identical tensors (the stream a job runs on never changes its result)."""
    import slim_amd.prune as P
    from slim_amd.layerwrapper import WrappedGPT
    from slim_amd.quantization.quantization import Quantizer

    def run(threads, trace):
        old = P.STREAM_THREADS, P.STREAM_TRACE
        P.STREAM_THREADS, P.STREAM_TRACE = threads, trace
        try:
            items = []
            for i, (n, k) in enumerate([(128, 256), (256, 256), (512, 384)]):     # three short dimensions: three chains, one per stream
                lin = torch.nn.Linear(k, n, bias=False, device=DEV, dtype=torch.bfloat16)
                lin.weight.data = synth_weight(n, k, 40 + i).to(DEV)
                wr = WrappedGPT(lin)
                wr.add_batch(synth_weight(64, k, 50 + i, sigma=1.0).to(DEV).reshape(1, 64, k), None)
                wr.scaler_row = ops.shift_zeros(wr.scaler_row)
                items.append((f"l{i}", lin, wr))
            P.compress_linears(items, torch.device(DEV), 2, 4, quantizer=Quantizer("weight", num_bits=4, slim_quant=True),
                               lora_rank=0.1, slim_lora=True, n_streams=3)
            torch.cuda.synchronize()
            return [(lin.weight.data.clone(), lin.lora_left.data.clone(), lin.lora_right.data.clone()) for _, lin, _ in items]
        finally:
            P.STREAM_THREADS, P.STREAM_TRACE = old

    base = run(True, False)
    for threads, trace in ((False, False), (True, True)):
        other = run(threads, trace)
        for a, b in zip(base, other):
            assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert len(P.LAST_STREAM_TRACE) == 3 and all(r["gpu_end_ms"] >= r["gpu_start_ms"] for r in P.LAST_STREAM_TRACE)
