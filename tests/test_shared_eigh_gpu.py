"""The fit in two halves around a shared, batched tridiagonalisation (slim_lowrank_fit_front / slim_tridiagonalize_f32_batched /
slim_eigh_tail_f32 / slim_lowrank_fit_back) against the one-call fit: same arithmetic, so identical factors."""
import pytest
import torch

from oracle import slim_oracle as O
from conftest import synth_weight, synth_scaler

pytestmark = pytest.mark.gpu


def test_split_fit_equals_single_call():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from slim_amd import ops
    from slim_amd.lora import FIT_ITERS
    r, over = 64, 64
    shapes = [(768, 512), (512, 768), (1024, 512)]
    Es, ctxs, refs, ss = [], [], [], []
    l = None
    for i, (N, K) in enumerate(shapes):
        W = synth_weight(N, K, 170 + i)
        sr = O.shift_zeros(synth_scaler(K, 180 + i))
        mask = O.prune_nm(O.saliency(W, sr), 2, 4)
        E = O.add_lora(W, mask, sr, 0.1, tile=None)["error_mat"].to(torch.bfloat16).cuda()
        Es.append((E, E.t().contiguous()))
        ss.append(torch.sqrt(sr).cuda())
        l = ops.fit_subspace(N, K, r, over)
    Mb = torch.empty(len(shapes), l, l, dtype=torch.float32, device="cuda")
    for i, (E, Et) in enumerate(Es):
        refs.append(ops.lowrank_fit(E, Et, r, over, FIT_ITERS, seed=0, s=ss[i]))
        ctxs.append(ops.lowrank_fit_front(E, Et, r, over, FIT_ITERS, Mb[i], seed=0))
    d, e, Q = ops.tridiagonalize_batched(Mb)
    for i in range(len(shapes)):
        got = ops.lowrank_fit_back(ctxs[i], d[i], e[i], Q[i], Mb[i], s=ss[i])
        torch.cuda.synchronize()
        assert int(got["info"][..., :-3].abs().sum()) == 0
        for k in ("U", "Ut", "L"):
            assert torch.equal(got[k], refs[i][k]), (i, k)
        torch.testing.assert_close(got["S"], refs[i]["S"], rtol=0, atol=0)


def test_batched_split_fit_equals_single_call():
    """Three same-shape fits in lockstep through the batched halves: every member equals its single-call fit."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from slim_amd import ops
    from slim_amd.lora import FIT_ITERS
    r, over, N, K, B = 64, 64, 768, 1024, 3
    Eg = torch.empty(B, N, K, dtype=torch.bfloat16, device="cuda")
    Etg = torch.empty(B, K, N, dtype=torch.bfloat16, device="cuda")
    ss, refs = [], []
    for i in range(B):
        W = synth_weight(N, K, 270 + i)
        sr = O.shift_zeros(synth_scaler(K, 280 + i))
        mask = O.prune_nm(O.saliency(W, sr), 2, 4)
        Eg[i] = O.add_lora(W, mask, sr, 0.1, tile=None)["error_mat"].to(torch.bfloat16).cuda()
        Etg[i] = Eg[i].t()
        ss.append(torch.sqrt(sr).cuda())
        refs.append(ops.lowrank_fit(Eg[i], Etg[i], r, over, FIT_ITERS, seed=0, s=ss[i]))
    l = ops.fit_subspace(N, K, r, over)
    Mb = torch.empty(B, l, l, dtype=torch.float32, device="cuda")
    ctx = ops.lowrank_fit_front(Eg, Etg, r, over, FIT_ITERS, Mb, seed=0)
    d, e, Q = ops.tridiagonalize_batched(Mb)
    got = ops.lowrank_fit_back(ctx, d, e, Q, Mb, s=torch.stack(ss))
    torch.cuda.synchronize()
    assert int(got["info"][..., :-3].abs().sum()) == 0
    for i in range(B):
        for k in ("U", "Ut", "L"):
            assert torch.equal(got[k][i], refs[i][k]), (i, k)
        torch.testing.assert_close(got["S"][i], refs[i]["S"], rtol=0, atol=0)


@pytest.mark.parametrize("outlier_gain", [1.0, 300.0])
def test_group_fits_equal_separate_fits(outlier_gain):
    """compress_linears with the same-input linears of a block fitted as one lockstep group, as separate chains around the
    shared eigen-solver, and with one eigen-solve per fit: identical layers.  With outlier channels in the shared input
    the grouped members deflate them before the lockstep fit."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import slim_amd.prune as P
    from slim_amd import ops
    from slim_amd.layerwrapper import WrappedGPT
    from slim_amd.quantization.quantization import Quantizer
    dev = torch.device("cuda", 0)

    import slim_amd.lora as LR

    def run(shared, grouped):
        # (the driver cuts the sketch products along their reduction only when fewer than three chains run side by side,
        # which changes rounding: switched off here so that all three modes do the same arithmetic)
        old = P.SHARED_EIGH, P.GROUP_FITS, LR.FIT_SKETCH_SPLIT
        P.SHARED_EIGH, P.GROUP_FITS, LR.FIT_SKETCH_SPLIT = shared, grouped, False
        try:
            X = synth_weight(256, 1024, 300, sigma=1.0)
            X[:, [5, 77, 900]] *= outlier_gain
            X2 = synth_weight(256, 512, 301, sigma=1.0)
            items = []
            for i, (n, k, x) in enumerate([(512, 1024, X), (512, 1024, X), (512, 1024, X), (1024, 512, X2), (768, 1024, X)]):
                lin = torch.nn.Linear(k, n, bias=False, device=dev, dtype=torch.bfloat16)
                lin.weight.data = synth_weight(n, k, 310 + i).to(dev)
                wr = WrappedGPT(lin)
                wr.add_batch(x.to(dev).reshape(1, *x.shape), None)
                wr.scaler_row = ops.shift_zeros(wr.scaler_row)
                items.append((f"l{i}", lin, wr))
            P.compress_linears(items, dev, 2, 4, quantizer=Quantizer("weight", num_bits=4, slim_quant=True),
                               lora_rank=0.1, slim_lora=True)
            torch.cuda.synchronize()
            if outlier_gain > 1:
                assert items[0][2].outliers is not None and items[0][2].outliers[1] == 3
            return [(lin.weight.data.clone(), lin.lora_left.data.clone(), lin.lora_right.data.clone()) for _, lin, _ in items]
        finally:
            P.SHARED_EIGH, P.GROUP_FITS, LR.FIT_SKETCH_SPLIT = old

    base = run(False, False)
    for shared, grouped in ((True, False), (True, True)):
        other = run(shared, grouped)
        for i, (a, b) in enumerate(zip(base, other)):
            assert all(torch.equal(x, y) for x, y in zip(a, b)), (shared, grouped, i)


def test_underestimated_conditioning_is_detected_and_the_fit_repeated():
    """A layer whose saliency scales spread further than the driver's estimate allows for: the fit reports the level it
    found (last status word), compress_linears repeats that layer's fit with it, and the result is as good as a fit that
    was sized generously from the start."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import slim_amd.prune as P
    import slim_amd.lora as LORA
    from slim_amd.layerwrapper import WrappedGPT
    from slim_amd.quantization.quantization import Quantizer
    N, K = 1024, 2048
    W = synth_weight(N, K, 310)
    g = torch.Generator().manual_seed(311)
    s = torch.exp(torch.randn(K, generator=g) * 0.5)
    s[torch.randperm(K, generator=g)[:200]] *= 9.0          # many channels 9x the median: below the split-off ratio
    sr = (s ** 2 * 2048).float()

    def run(cond_e, hints=None):
        calls = []
        lin = torch.nn.Linear(K, N, bias=False).to(torch.bfloat16).cuda()
        lin.weight.data = W.cuda()
        wrapped = WrappedGPT(lin)
        wrapped.scaler_row = sr.cuda()
        wrapped.nsamples = 8
        P.block_stats([wrapped])
        real, real_steps = P.compress_linears_unchecked, LORA.fit_ns_steps
        # (what is recorded per pass: the level the driver asks the repeated fits to be sized for, 0 on the first pass)
        P.compress_linears_unchecked = lambda items, *a, **k: (calls.append(max(w.fit_cond_floor for _, _, w in items)),
                                                               real(items, *a, **k))[1]
        if cond_e is not None:
            LORA.fit_ns_steps = lambda spread, hint=0, floor=0: max(cond_e, floor)
        try:
            P.compress_linears([("lin", lin, wrapped)], torch.device("cuda", 0), 2, 4, quantizer=Quantizer("weight", num_bits=4, slim_quant=True),
                               lora_rank=0.1, slim_lora=True, quantize_lora=True, lora_tile_size=128, level_hints=hints)
        finally:
            P.compress_linears_unchecked, LORA.fit_ns_steps = real, real_steps
        eff = lin.weight.data.float() + (lin.lora_right.data.float().t() @ lin.lora_left.data.float().t()) / float(lin.lora_rank)
        err = torch.norm((W.cuda().float() - eff) * torch.sqrt(sr).cuda().reshape(1, -1)).item()
        return err, calls, int(lin.lora_fit_info[-1])
    err_wide, calls_wide, word_wide = run(6)
    err_low, calls_low, word_low = run(1)
    print(f"weighted error: sized generously {err_wide:.4f} ({len(calls_wide)} pass), underestimated {err_low:.4f} "
          f"({len(calls_low)} passes, repeated at level {calls_low[-1]})")
    assert len(calls_wide) == 1 and word_wide == 0
    assert len(calls_low) == 2 and calls_low[1] >= 2           # detected, repeated with the level the fit reported
    assert err_low <= err_wide * (1 + 1e-3)

    # the same through a hint that is too low (a previous block with a much narrower spectrum): detected, repeated without
    # the hint, and the dictionary then holds what this layer measured
    hints = {("lin", (N, K)): 2}
    err_hint, calls_hint, _ = run(None, hints)
    print(f"too low a hint: {err_hint:.4f} ({len(calls_hint)} passes), hint left behind {hints}")
    assert len(calls_hint) == 2 and hints[("lin", (N, K))] > 2
    assert err_hint <= err_wide * (1 + 1e-3)
    # and with the hint it left behind: one pass, as good
    err_next, calls_next, word_next = run(None, hints)
    assert len(calls_next) == 1 and word_next == 0 and err_next <= err_wide * (1 + 1e-3)
    # a hint whose cut sits too high for this layer (a previous block with a spectrum that ends much higher): the filtered
    # iteration damps directions it should keep, the fit notices (its own cut lies below the interval's end), is repeated
    # unfiltered, and leaves its own measurement behind
    good = hints[("lin", (N, K))]
    assert LORA.filter_plan([good]) is not None
    too_high = (good & 0xffff) | (min(int((good >> 16) * 1.6), 65535) << 16)
    hints2 = {("lin", (N, K)): too_high}
    err_f, calls_f, _ = run(None, hints2)
    print(f"filter interval too high: {err_f:.4f} ({len(calls_f)} passes), ratio left behind {(hints2[('lin', (N, K))] >> 16) / 2048:.3f} "
          f"(was given {(too_high >> 16) / 2048:.3f}, measured before {(good >> 16) / 2048:.3f})")
    assert len(calls_f) == 2 and calls_f[1] >= 1
    assert err_f <= err_wide * (1 + 1e-3)
    assert abs((hints2[("lin", (N, K))] >> 16) / (good >> 16) - 1) < 0.1
