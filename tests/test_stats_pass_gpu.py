"""The statistics pass of a block (slim_amd.prune.gather_block_statistics): stopping the block's forward at the input of its
last linear leaves every linear's statistics exactly as the plain pass computes them."""
import pytest
import torch

pytestmark = pytest.mark.gpu


class Block(torch.nn.Module):
    def __init__(self, twice=False):
        super().__init__()
        self.a = torch.nn.Linear(64, 128, bias=False)
        self.b = torch.nn.Linear(64, 128, bias=False)
        self.c = torch.nn.Linear(128, 64, bias=False)
        self.twice = twice

    def forward(self, x):
        h = torch.relu(self.a(x)) * self.b(x)
        y = self.c(h)
        if self.twice:
            y = y + self.c(torch.relu(h))
        return x + y


@pytest.mark.parametrize("twice", [False, True])
def test_early_exit_keeps_statistics(twice):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import slim_amd.prune as P
    from slim_amd.layerwrapper import WrappedGPT
    from slim_amd.utils import find_layers
    torch.manual_seed(3)
    layer = Block(twice).to(torch.bfloat16).cuda()
    inps = torch.randn(6, 16, 64, device="cuda").to(torch.bfloat16)
    results = {}
    for early in (False, True):
        subset = find_layers(layer)
        wrapped = {n: WrappedGPT(subset[n]) for n in subset}
        outs = torch.zeros_like(inps)
        saved, P.STATS_EARLY_EXIT = P.STATS_EARLY_EXIT, early
        try:
            P.gather_block_statistics(layer, subset, wrapped, inps, outs, {}, 2)
        finally:
            P.STATS_EARLY_EXIT = saved
        torch.cuda.synchronize()
        assert all(len(m._forward_hooks) == 0 and len(m._forward_pre_hooks) == 0 for m in subset.values())
        results[early] = ({n: (w.scaler_row.clone(), w.nsamples) for n, w in wrapped.items()}, outs)
    plain, fast = results[False], results[True]
    for n in plain[0]:
        assert plain[0][n][1] == fast[0][n][1] == (12 if (twice and n == "c") else 6)
        assert torch.equal(plain[0][n][0], fast[0][n][0]), n
    assert torch.equal(plain[1][:2], fast[1][:2])                  # the first batch runs whole either way
    if twice:
        assert torch.equal(plain[1], fast[1])                       # a linear that runs twice: no early exit
    else:
        assert float(fast[1][2:].abs().max()) == 0.0                # later batches stop before the block's output


class Flipping(torch.nn.Module):
    """Two independent branches whose order flips after the first call."""

    def __init__(self):
        super().__init__()
        self.p = torch.nn.Linear(64, 64, bias=False)
        self.q = torch.nn.Linear(64, 64, bias=False)
        self.calls = 0

    def forward(self, x):
        self.calls += 1
        if self.calls % 3 == 1:
            return self.p(x) + self.q(torch.relu(x))
        b = self.q(torch.relu(x))
        return self.p(x) + b


def test_early_exit_survives_a_changed_order():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import slim_amd.prune as P
    from slim_amd.layerwrapper import WrappedGPT
    from slim_amd.utils import find_layers
    torch.manual_seed(4)
    layer = Flipping().to(torch.bfloat16).cuda()
    inps = torch.randn(6, 16, 64, device="cuda").to(torch.bfloat16)
    stats = {}
    for early in (False, True):
        layer.calls = 0
        subset = find_layers(layer)
        wrapped = {n: WrappedGPT(subset[n]) for n in subset}
        outs = torch.zeros_like(inps)
        saved, P.STATS_EARLY_EXIT = P.STATS_EARLY_EXIT, early
        try:
            P.gather_block_statistics(layer, subset, wrapped, inps, outs, {}, 2)
        finally:
            P.STATS_EARLY_EXIT = saved
        torch.cuda.synchronize()
        stats[early] = {n: (w.scaler_row.clone(), w.nsamples) for n, w in wrapped.items()}
    for n in stats[False]:
        assert stats[False][n][1] == stats[True][n][1] == 6
        assert torch.equal(stats[False][n][0], stats[True][n][0]), n
