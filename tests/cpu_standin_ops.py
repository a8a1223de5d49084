"""Test infrastructure, not product code: CPU stand-ins (torch + the oracle) for the slim_amd.ops functions that
slim_amd/sharded.py calls, with the same signatures and the same rounding points (bf16 operands, fp32 accumulation), so
that the orchestration of the sharded path -- which blocks a rank works on, what is reduced, broadcast and gathered, how
the factors are assembled -- runs over a gloo group on machines without a GPU.  The iterative pieces are replaced by
their exact counterparts: the subspace iteration by the leading eigenvectors of the Gram matrix, the eigen-solver by
torch.linalg.eigh, Newton-Schulz by the polar factor."""
import torch

from oracle import slim_oracle as O

BF16 = torch.bfloat16


def sqrt(v):
    return torch.sqrt(v)


def shift_zeros(v):
    return O.shift_zeros(v)


def transpose(x, out=None):
    return x.t().contiguous()


def gram_accum(x, H, nsamples_before, batch):
    """sparsegpt.py:36-39 in place (bf16 products are exact in fp32; fp32 accumulation)."""
    new, _ = O.add_batch_gram(H, nsamples_before, x.reshape(batch, -1, x.shape[-1]))
    H.copy_(new)
    return H


def scaler_row_accum(x, scaler_row, nsamples_before, n_chunks=1, batch_per_chunk=1):
    """layerwrapper.py:25-41 in place, one update per chunk."""
    n = nsamples_before
    for chunk in x.reshape(n_chunks, -1, x.shape[-1]):
        new, n = O.add_batch_scaler_row(scaler_row, n, chunk.reshape(batch_per_chunk, -1, x.shape[-1]))
        scaler_row.copy_(new)
    return scaler_row


def mask_nm(W, s, n, m):
    return O.prune_nm(W.abs() * s.reshape(1, -1), n, m)               # slim/utils.py:75-88 on |W| * sqrt(scaler_row)


def hist_abs(W, hist=None):
    """Counts per |bf16| bit pattern (15 bits), accumulated into hist when given."""
    bits = (W.contiguous().view(torch.int16).to(torch.int32) & 0x7fff).flatten()
    h = torch.bincount(bits, minlength=32768).to(torch.int32)
    if hist is None:
        return h
    hist += h
    return hist


def _values(hist):
    """The multiset of |W| the histogram stands for, as a bf16 vector."""
    patterns = torch.arange(32768, dtype=torch.int32).to(torch.int16).view(BF16)
    return torch.repeat_interleave(patterns, hist.to(torch.int64))


def quant_search(hist, num_bins, num_bits):
    alpha = O.find_optimal_quantization_cap(_values(hist), num_bits, num_bins)        # quantization.py:34-62
    sf = (2 ** (num_bits - 1) - 1) / alpha
    return alpha, torch.tensor(0), sf, None


def absmax_scale(hist, num_bits):
    alpha = _values(hist).max().float()
    return alpha, (2 ** (num_bits - 1) - 1) / alpha


def _codes(W, sf, num_bits):
    max_q = 2 ** (num_bits - 1) - 1
    return torch.clamp(torch.round((W * sf.reshape(())).float()), -max_q - 1, max_q)   # quantization.py:194-200


def fake_quant_pt(W, sf, num_bits, want_codes=False):
    return (_codes(W, sf, num_bits) / sf.reshape(())).to(BF16)


def error_matrix(W, mask, s, sf, num_bits, want_f32=False, want_bf16=True, out=None, col_flags=None, deq_bf16=False):
    """slim/lora.py:146-156: E = W s - mask0(deq(q(W)) s)."""
    assert col_flags is None
    sc = s.reshape(1, -1) if s is not None else 1.0
    deq = _codes(W, sf, num_bits) / sf.reshape(())
    if deq_bf16:
        deq = deq.to(BF16).float()
    new_w = deq * sc
    new_w[mask] = 0
    E = W * sc - new_w
    return (E.float() if want_f32 else None), E.to(BF16)


def gemm_nt(A, Bt, alpha=1.0, beta=0.0, C32=None, out_f32=False, out_bf16=False, out_t=False, C16=None):
    C = alpha * (A.float() @ Bt.float().t())
    if C16 is not None:
        C16.copy_(C)
        out_bf16 = True
    return (C if out_f32 else None), (C.to(BF16) if out_bf16 else None), (C.to(BF16).t().contiguous() if out_t else None)


def orthonormalize(X, Xt=None, ns_quintic=2, ns_cubic=3, cholesky_qr=False):
    """The polar factor X (X^T X)^(-1/2), which Newton-Schulz steps converge to; zero columns stay zero."""
    Xf = X.double()
    w, V = torch.linalg.eigh(Xf.t() @ Xf)
    inv = torch.where(w > 1e-9 * w.max(), w.clamp_min(1e-300).rsqrt(), torch.zeros_like(w))
    Q = (Xf @ (V * inv) @ V.t()).to(BF16)
    return Q, Q.t().contiguous(), torch.zeros(64, dtype=torch.int32)


def outlier_basis(Et, cols, t):
    """Q_O = E_O D^-1 R^-1 from the Cholesky factor of the normalised columns' Gram matrix, as the library forms it."""
    tp = cols.numel()
    EO = Et.double()[cols[:t].long()].t()                     # [N, t]
    dn = EO.norm(dim=0)
    R = torch.linalg.cholesky((EO.t() @ EO) / (dn[:, None] * dn[None, :]), upper=True)
    Q = torch.zeros(EO.shape[0], tp, dtype=torch.float64)
    Q[:, :t] = (EO / dn) @ torch.linalg.inv(R)
    Q = Q.to(BF16)
    return Q, Q.t().contiguous()


def gramfit_iterate(C16, l, n_iter, seed=0, flags=0, cond_e=0):
    w, V = torch.linalg.eigh(C16.double())
    P = V[:, -l:].to(BF16).contiguous()
    return P, P.t().contiguous(), torch.zeros(64, dtype=torch.int32)


def gramfit_iterate_batched(C16, l, n_iter, seed=0, flags=0, cond_e=0):
    outs = [gramfit_iterate(C, l, n_iter, seed, flags, cond_e) for C in C16]
    B = len(outs)
    return (torch.stack([o[0] for o in outs]), torch.stack([o[1] for o in outs]), torch.zeros(B, 64, dtype=torch.int32),
            torch.zeros(B, 4))


def fit_subspace(N, K, r, oversample):
    l = min(r + oversample, N, K)
    return l - l % 8


def set_fit_filter(ratios, bits):
    pass            # (the stand-in iteration is exact: nothing to filter)


def gramfit_check(evals, r, cond_e, scal, info):
    return info     # (exact stand-ins: no conditioning to report)


def ritz_split_supported(l, drop_max):
    return False    # (the stand-in Rayleigh-Ritz stage is the exact eigen-decomposition below)


def gramfit_project(X, Pt, M32=None):
    Z = (X.float() @ Pt.float().t()).to(BF16)
    M = Z.float().t() @ Z.float()
    if M32 is not None:
        M32.copy_(M)
        M = M32
    return Z, Z.t().contiguous(), M


def lora_adapters(U, L, rank, prescale=True):
    c = float(rank) ** 0.5 if prescale else 1.0
    return (L[:, :rank].float() * c).to(BF16).contiguous(), (U[:, :rank].float() * c).to(BF16).t().contiguous()


def outlier_project(Et, UOt):
    return (Et.float() @ UOt.float().t()).to(BF16)


def deflate_apply(E, UO, Ct, out=None):
    E2 = (E - (UO.float() @ Ct.float().t()).to(BF16)).to(BF16)
    return E2, E2.t().contiguous()


def lora_assemble(UO, Ct, s, fitU, fitL, t, rank, r8):
    """[exact part | the fit's factors | 0] (slim_lora_assemble)."""
    N, K = UO.shape[0], fitL.shape[0]
    U = torch.zeros(N, r8, dtype=BF16)
    L = torch.zeros(K, r8, dtype=BF16)
    U[:, :t] = UO[:, :t]
    L[:, :t] = Ct[:, :t] if s is None else (Ct[:, :t] / s.to(BF16).reshape(-1, 1)).to(BF16)
    U[:, t:rank] = fitU[:, :rank - t]
    L[:, t:rank] = fitL[:, :rank - t]
    return U, U.t().contiguous(), L


def tridiagonalize_batched(A):
    """Stand-in for both stages of the eigen-solver: d = eigenvalues (ascending), Q = eigenvectors."""
    w, V = torch.linalg.eigh(A.double())
    return w.float().contiguous(), torch.zeros_like(w, dtype=torch.float32), V.float().contiguous()


def eigh_tail(d, e, Q, vecs, top):
    vecs.copy_(Q.transpose(-1, -2))          # row n-1-i = eigenvector of the i-th largest eigenvalue
    return torch.zeros(d.shape[0], dtype=torch.int32)


def scale_rows(L, s):
    L.copy_(L / s.to(BF16).reshape(-1, 1))
    return L


def gramfit_factors(right, P, Z, vecs, evals, r, s=None, want_dm_t=False, want_rows_t=False, info=None, cond_e=0):
    l = vecs.shape[0]
    W = vecs.flip(0)[:r].t().contiguous()                    # [l, r], column i = eigenvector of the i-th largest
    S = evals.flip(0)[:r].clamp_min(0).sqrt()
    if right:
        dm = (P.float() @ (W * S).to(BF16).float()).to(BF16)
        rows = (Z.float() @ (W / S.clamp_min(1e-30)).to(BF16).float()).to(BF16)
        if s is not None:
            scale_rows(dm, s)
    else:
        dm = (P.float() @ W.to(BF16).float()).to(BF16)
        rows = (Z.float() @ W.to(BF16).float()).to(BF16)
        if s is not None:
            scale_rows(rows, s)
    return dm, None, rows, None, S


def lowrank_residual(U, L, W, mask, want_lr=False, out=None):
    LR = (U.float() @ L.float().t()).to(BF16)                 # slim/lora.py:199-201
    Wn = (W - LR).to(BF16)
    if mask is not None:
        Wn[mask] = 0
    return Wn
