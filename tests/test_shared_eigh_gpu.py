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
        assert int(got["info"].abs().sum()) == 0
        for k in ("U", "Ut", "L"):
            assert torch.equal(got[k], refs[i][k]), (i, k)
        torch.testing.assert_close(got["S"], refs[i]["S"], rtol=0, atol=0)
