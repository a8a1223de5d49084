"""The in-library symmetric eigen-solver (one launch per column of the tridiagonalisation, the tridiagonal problem by
multisection and twisted factorisations in double precision, one fp32 GEMM) against numpy.linalg.eigh in float64:
eigenvalues, residuals and orthonormality of the leading eigenvectors."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _spectrum(kind, n, rng):
    if kind == "gram":          # Gram matrix of a tall random matrix: the Rayleigh-Ritz matrices of the fit look like this
        X = rng.standard_normal((4 * n, n)) * np.logspace(0, -2, n)
        return X.T @ X
    if kind == "clustered":     # repeated eigenvalues
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        w = np.repeat(np.arange(1, n // 4 + 1), 4)[:n].astype(np.float64)
        return (Q * w) @ Q.T
    if kind == "indefinite":
        A = rng.standard_normal((n, n))
        return (A + A.T) / 2
    if kind == "diagonal":      # every reflector is the identity
        return np.diag(np.arange(1, n + 1, dtype=np.float64))
    if kind == "lowrank":       # numerically rank deficient
        X = rng.standard_normal((n, 5))
        return X @ X.T
    raise ValueError(kind)


@pytest.mark.parametrize("n,kind", [(4, "indefinite"), (8, "gram"), (64, "indefinite"), (100, "clustered"), (256, "gram"),
                                    (260, "indefinite"), (640, "gram"), (640, "lowrank"), (64, "diagonal"), (776, "clustered"),
                                    (1032, "gram"), (1400, "indefinite")])
def test_eigh_matches_numpy(n, kind):
    from slim_amd import ops
    rng = np.random.default_rng(n)
    A = _spectrum(kind, n, rng)
    A = (A + A.T) / 2
    A32 = A.astype(np.float32)
    w_ref = np.linalg.eigh(A32.astype(np.float64))[0]
    top = max(1, (n * 4) // 5)
    evals, vecs, info = ops.eigh(torch.from_numpy(A32).cuda(), top=top)
    torch.cuda.synchronize()
    assert int(info.item()) == 0
    w = evals.cpu().numpy().astype(np.float64)
    scale = np.abs(w_ref).max()
    # fp32 backward-stable solver: eigenvalue error a modest multiple of eps * ||A||
    assert np.abs(w - w_ref).max() <= 2e-5 * scale, np.abs(w - w_ref).max() / scale
    Z = vecs.cpu().numpy().astype(np.float64)[n - top:]          # rows = eigenvectors, ascending
    lam = w[n - top:]
    resid = np.abs(Z @ A32.astype(np.float64) - lam[:, None] * Z).max()
    assert resid <= 5e-5 * scale, resid / scale
    orth = np.abs(Z @ Z.T - np.eye(top)).max()
    assert orth <= 2e-4, orth


@pytest.mark.parametrize("n", [4, 64, 260, 640])
def test_tridiagonalize_is_an_orthogonal_similarity(n):
    from slim_amd import ops
    rng = np.random.default_rng(n + 1)
    A = rng.standard_normal((n, n))
    A = ((A + A.T) / 2).astype(np.float32)
    d, e, Q = (t.cpu().numpy().astype(np.float64) for t in ops.tridiagonalize(torch.from_numpy(A).cuda()))
    T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    scale = np.abs(A).max() * n ** 0.5
    assert np.abs(Q.T @ Q - np.eye(n)).max() <= 1e-5 * n ** 0.5
    assert np.abs(Q.T @ A.astype(np.float64) @ Q - T).max() <= 2e-6 * scale * n ** 0.5


def test_eigh_rejects_bad_arguments():
    from slim_amd import ops
    with pytest.raises(Exception):
        ops.eigh(torch.zeros(6, 6, device="cuda"))           # n not a multiple of 4
    with pytest.raises(Exception):
        ops.eigh(torch.zeros(8, 8, device="cuda"), top=9)
