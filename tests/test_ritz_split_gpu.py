"""slim_ritz_split_f32_batched (the fit's Rayleigh-Ritz step as a Chebyshev-filtered search for the small subspace to
drop, no eigen-decomposition) against an exact eigen-decomposition of the same matrices: the basis is orthonormal, and the
energy it keeps is within a stated fraction of the optimum -- on synthetic spectra and on the Rayleigh-Ritz matrices of
full-size SLiM error matrices, where the loss is priced in the fit's own currency (relative reconstruction error)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from slim_amd import ops as _ops
    return _ops


def _sym(w, seed):
    g = torch.Generator().manual_seed(seed)
    l = len(w)
    Q, _ = torch.linalg.qr(torch.randn(l, l, generator=g, dtype=torch.float64))
    return ((Q * w) @ Q.T).float()


def _check(ops, M, top, keep, tol_trace):
    """M [B, l, l] on the GPU.  Returns the per-problem excess of the dropped trace over the optimum (relative)."""
    B, l, _ = M.shape
    vecs, evals, info = ops.ritz_split(M, top, keep)
    assert int(info.abs().sum()) == 0
    out = []
    for b in range(B):
        k = keep[b]
        W = vecs[b, l - k:].double()                       # [k, l] rows = basis vectors
        G = W @ W.T
        assert float((G - torch.eye(k, dtype=torch.float64, device=DEV)).abs().max()) < 2e-5, b
        if k < top:
            assert float(vecs[b, l - top:l - k].abs().max()) == 0.0
        Md = M[b].double()
        w = torch.linalg.eigvalsh(Md)
        kept = float(torch.trace(W @ Md @ W.T))
        opt_drop = float(w[:l - k].sum())
        drop = float(w.sum()) - kept
        rq = torch.einsum("ij,jk,ik->i", W, Md, W)
        got = evals[b, l - k:].double()
        # Rayleigh quotients (slot l-1 holds an estimate of the largest eigenvalue instead)
        assert torch.allclose(got[:-1], rq[:-1], rtol=2e-4, atol=1e-6 * float(w[-1]))
        assert float(evals[b, l - 1]) >= 0.9 * float(w[-1]) and float(evals[b, l - 1]) <= 1.35 * float(w[-1])
        ex = drop / max(opt_drop, 1e-30) - 1 if opt_drop > 1e-6 * float(w.sum()) else drop / float(w.sum())
        assert ex <= tol_trace, (b, ex)
        assert ex >= -1e-4
        out.append(ex)
    return out


@pytest.mark.parametrize("name,w,keep,tol", [
    ("flat 2x, l 320", torch.linspace(1, 2, 320), 256, 2e-3),
    ("flat 2x, odd keep", torch.linspace(1, 2, 320), 251, 2e-3),
    ("gap", torch.cat([torch.full((64,), 1.0), torch.linspace(5, 6, 256)]), 256, 1e-5),
    ("geometric 100x", torch.logspace(0, 2, 320, dtype=torch.float64), 256, 5e-2),
    ("rank deficient", torch.cat([torch.zeros(100), torch.linspace(1, 3, 220)]), 256, 1e-5),
    ("small, most dropped", torch.logspace(0, 1.5, 96, dtype=torch.float64), 25, 1e-4),
    ("l 576", torch.linspace(1, 3.5, 576), 508, 3e-3),
    ("nothing dropped", torch.linspace(1, 2, 64), 64, 1e-6),
])
def test_split_on_synthetic_spectra(ops, name, w, keep, tol):
    l = len(w)
    top = min(l, -(-keep // 8) * 8)
    M = _sym(w.double(), 5).to(DEV).unsqueeze(0).contiguous()
    ex = _check(ops, M, top, [keep], tol)
    print(name, "excess of the dropped trace", ex)


def test_split_batch_with_different_keeps(ops):
    w = torch.linspace(1, 3, 320).double()
    M = torch.stack([_sym(w, 1), _sym(w * 2, 2), _sym(w.flip(0) * 0.5, 3)]).to(DEV)
    one = [ops.ritz_split(M[b:b + 1].contiguous(), 256, [k])[0] for b, k in enumerate([256, 250, 253])]
    vecs, _, _ = ops.ritz_split(M, 256, [256, 250, 253])
    for b in range(3):
        assert torch.equal(vecs[b, 320 - 256:], one[b][0, 320 - 256:]), b           # a member of a batch equals the same problem alone
    _check(ops, M, 256, [256, 250, 253], 2e-3)


@pytest.mark.parametrize("N,K,rank", [(4096, 4096, 512), (4096, 11008, 512), (768, 768, 256), (3072, 768, 256), (1024, 8192, 256)])
def test_split_on_full_size_rayleigh_ritz_matrices(ops, N, K, rank):
    """The matrix the fit's front half produces on a full-size error matrix: what the split loses against the exact
    eigen-decomposition, as a fraction of the fit's reconstruction error, stays below 4e-4 (the fit's budget is 1e-3)."""
    import test_fullsize_gpu as T
    from slim_amd.lora import fit_plan, outlier_columns, deflated_pair, fit_iters, _fit_kwargs
    E, sr = T._error_matrix(ops, N, K, 100 + N % 97 + K % 89)
    outl, spread = outlier_columns(sr)
    rf, over, l, t = fit_plan(N, K, rank, outl)
    Et = ops.transpose(E)
    if t:
        E, Et, _, _, _ = deflated_pair(E, Et, outl)
    M = torch.empty(1, l, l, dtype=torch.float32, device=DEV)
    ops.lowrank_fit_front(E, Et, rf, over, fit_iters(l), M[0], **_fit_kwargs(spread))
    keep = rank - t
    total = float((E.float() ** 2).sum())
    w = torch.linalg.eigvalsh(M[0].double())
    err2 = total - float(w[l - keep:].sum())
    vecs, evals, info = ops.ritz_split(M, rf, [keep])
    W = vecs[0, l - keep:].double()
    kept = float(torch.trace(W @ M[0].double() @ W.T))
    loss = (float(w[l - keep:].sum()) - kept) / (2 * err2)
    print(f"{N}x{K} r={rank} l={l} keep={keep}: relative reconstruction error lost by the split {loss:.2e}; "
          f"eigenvalue spread {float(w[-1] / w[0]):.1f}")
    assert 0 <= loss + 1e-7 and loss <= 4e-4, loss
    _check(ops, M, rf, [keep], 1e-2)
