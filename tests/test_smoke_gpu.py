"""The entry point's smoke check as a test: a small layer (512 x 1024, rank 128, subspace 192 of 512), where the weakest
directions of the last iterate are far less well conditioned than on full-size layers -- the case that exposed a final
orthonormalisation that was only good enough for large layers."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


@pytest.mark.gpu
def test_smoke_entry_point():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import __graft_entry__
    __graft_entry__.smoke()


@pytest.mark.gpu
@pytest.mark.parametrize("knobs", [dict(sketch_split=False), dict(library_eigh=True)])
def test_small_layer_fit_is_insensitive_to_the_solver_knobs(knobs):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import slim_oracle as O
    from conftest import synth_weight, synth_scaler
    from slim_amd import ops
    from slim_amd.lora import FIT_ITERS
    N, K, r = 512, 1024, 128
    W = synth_weight(N, K, 7)
    sr = O.shift_zeros(synth_scaler(K, 8))
    mask = O.prune_nm(O.saliency(W, sr), 2, 4)
    E = O.add_lora(W, mask, sr, 0.1, tile=128)["error_mat"].to(torch.bfloat16)
    Eb, Ef = E.cuda(), E.float()
    Et = Eb.t().contiguous()
    errs = []
    for kw in (dict(), knobs):
        f = ops.lowrank_fit(Eb, Et, r, 64, FIT_ITERS, seed=0, ns_steps=3, **kw)
        errs.append(torch.norm(Ef - f["U"].cpu().float() @ f["L"].cpu().float().t()).item())
    assert abs(errs[0] / errs[1] - 1) <= 2e-3, errs
