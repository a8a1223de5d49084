"""Tensor-level wrappers over the C ABI (include/slim_hip.h).

PyTorch is used for device memory and streams only: every function allocates its outputs
with torch, passes raw device pointers and the current HIP stream to libslim_hip.so, and
returns torch tensors.  Inputs must live on a ROCm device; nothing here computes on the CPU.
"""
import os

import torch

from . import _lib

BF16 = torch.bfloat16


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _chk(t, dtype=None, name="tensor"):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be on the GPU: slim_amd has no CPU path (got device {t.device})")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# ---- calibration statistics -------------------------------------------------------------
def scaler_row_accum(x, scaler_row, nsamples_before, n_chunks, batch_per_chunk):
    """x: bf16 [n_chunks * rows, K]; scaler_row fp32 [K] updated in place (n_chunks add_batch calls)."""
    x = _chk(x, BF16, "x")
    _chk(scaler_row, torch.float32, "scaler_row")
    K = x.shape[-1]
    x2 = x.reshape(-1, K)
    rows = x2.shape[0] // n_chunks
    assert rows * n_chunks == x2.shape[0]
    nb = _lib.query("slim_colsumsq_workspace_bytes", n_chunks, rows, K)
    ws = _ws(nb, x.device)
    _lib.call("slim_scaler_row_accum", _p(x2), n_chunks, rows, K, K, batch_per_chunk, _p(scaler_row),
              int(nsamples_before), _p(ws), ws.numel(), _stream())
    return scaler_row


def colsumsq(x, n_chunks):
    """Per-chunk column sums of squares of x (bf16 [n_chunks * rows, K]) -> fp32 [n_chunks, K]."""
    x = _chk(x, BF16, "x")
    K = x.shape[-1]
    x2 = x.reshape(-1, K)
    rows = x2.shape[0] // n_chunks
    assert rows * n_chunks == x2.shape[0]
    out = torch.empty(n_chunks, K, dtype=torch.float32, device=x.device)
    ws = _ws(_lib.query("slim_colsumsq_workspace_bytes", n_chunks, rows, K), x.device)
    _lib.call("slim_colsumsq", _p(x2), n_chunks, rows, K, K, _p(out), _p(ws), ws.numel(), _stream())
    return out


def scaler_row_fold(scaler_row, sumsq, nsamples_before, batch_per_chunk=1):
    """Apply the scaler_row recurrence to per-chunk sums (fp32 [n_chunks, K]), in place."""
    _chk(scaler_row, torch.float32, "scaler_row")
    sumsq = _chk(sumsq, torch.float32, "sumsq")
    n_chunks, K = sumsq.shape
    _lib.call("slim_scaler_row_fold", _p(scaler_row), K, _p(sumsq), n_chunks, batch_per_chunk, int(nsamples_before), _stream())
    return scaler_row


def gram_accum(x, H, nsamples_before, batch):
    x = _chk(x, BF16, "x")
    _chk(H, torch.float32, "H")
    K = x.shape[-1]
    x2 = x.reshape(-1, K)
    ws = _ws(_lib.query("slim_gram_workspace_bytes", x2.shape[0], K), x.device)
    _lib.call("slim_gram_accum", _p(x2), x2.shape[0], K, K, _p(H), int(nsamples_before), int(batch), _p(ws),
              ws.numel(), _stream())
    return H


def shift_zeros(v):
    v = _chk(v, torch.float32, "x")
    out = torch.empty_like(v)
    _lib.call("slim_shift_zeros", _p(v), _p(out), v.numel(), _stream())
    return out


def sqrt(v):
    v = _chk(v, torch.float32, "x")
    out = torch.empty_like(v)
    _lib.call("slim_sqrt_f32", _p(v), _p(out), v.numel(), _stream())
    return out


def any_zero_flag(v):
    """Device int32 scalar: 1 if any element of v is zero (no host synchronisation)."""
    v = _chk(v, torch.float32, "x")
    flag = torch.empty(1, dtype=torch.int32, device=v.device)
    _lib.call("slim_any_zero_f32", _p(v), v.numel(), _p(flag), _stream())
    return flag[0]


def any_zero(v):
    return bool(any_zero_flag(v).item())


def outlier_stats(scaler_row, top, out_vals=None, out_idx=None):
    """(vals fp32 [2 + top], idx int32 [top]) of slim_outlier_stats: any-zero flag, median of sqrt(scaler_row), the `top`
    largest sqrt values in decreasing order and their indices.  out_vals / out_idx: rows of preallocated tensors."""
    v = _chk(scaler_row, torch.float32, "scaler_row")
    if out_vals is None:
        out_vals = torch.empty(2 + top, dtype=torch.float32, device=v.device)
        out_idx = torch.empty(top, dtype=torch.int32, device=v.device)
    _lib.call("slim_outlier_stats", _p(v), v.numel(), int(top), _p(out_vals), _p(out_idx), _stream())
    return out_vals, out_idx


# ---- N:M mask --------------------------------------------------------------------------------
def mask_nm(W, s, n, m):
    """Fused saliency |W| * s and N:M selection; returns a bool [N, K] tensor (True = pruned)."""
    W = _chk(W, BF16, "W")
    if s is not None:
        s = _chk(s, torch.float32, "s")
        assert s.numel() == W.shape[1]
    mask = torch.empty(W.shape, dtype=torch.bool, device=W.device)
    _lib.call("slim_mask_nm", _p(W), _p(s), W.shape[0], W.shape[1], n, m, _p(mask), _stream())
    return mask


def mask_unstructured(W, s, k_prune):
    """Fused saliency |W| * s and per-row selection of the k_prune smallest (True = pruned, ties to the lower index)."""
    W = _chk(W, BF16, "W")
    s = _chk(s, torch.float32, "s")
    assert s.numel() == W.shape[1]
    mask = torch.empty(W.shape, dtype=torch.bool, device=W.device)
    _lib.call("slim_mask_unstructured", _p(W), _p(s), W.shape[0], W.shape[1], int(k_prune), _p(mask), _stream())
    return mask


def prune_unstructured_f32(metric, k_prune):
    metric = _chk(metric, torch.float32, "metric")
    mask = torch.empty(metric.shape, dtype=torch.bool, device=metric.device)
    _lib.call("slim_prune_unstructured_f32", _p(metric), metric.shape[0], metric.shape[1], int(k_prune), _p(mask), _stream())
    return mask


def prune_nm_f32(metric, n, m):
    metric = _chk(metric, torch.float32, "metric")
    mask = torch.empty(metric.shape, dtype=torch.bool, device=metric.device)
    _lib.call("slim_prune_nm_f32", _p(metric), metric.shape[0], metric.shape[1], n, m, _p(mask), _stream())
    return mask


# ---- SLiM-Quant ------------------------------------------------------------------------------
def hist_abs(W, hist=None):
    W = _chk(W, BF16, "W")
    acc = hist is not None
    if hist is None:
        hist = torch.empty(32768, dtype=torch.int32, device=W.device)
    _lib.call("slim_hist_abs_bf16", _p(W), W.numel(), _p(hist), int(acc), _stream())
    return hist


def quant_search(hist, num_bins, num_bits):
    """Returns (alpha 0-d fp32, idx 0-d int32, sf 0-d fp32) device tensors."""
    dev = hist.device
    out = torch.empty(2, dtype=torch.float32, device=dev)
    idx = torch.empty(1, dtype=torch.int32, device=dev)
    ws = _ws(_lib.query("slim_quant_search_workspace_bytes", num_bins), dev)
    _lib.call("slim_quant_search", _p(hist), num_bins, num_bits, _p(out), _p(idx), _p(out[1:]), _p(ws), ws.numel(),
              _stream())
    return out[0], idx[0], out[1], ws


def fmod_probe(a, b):
    """(the library's exact remainder of a / b, fmodf(a, b)) element-wise for fp32 a >= 0, b > 0 (slim_fmod_probe_f32)."""
    a, b = _chk(a, torch.float32, "a"), _chk(b, torch.float32, "b")
    fast, ref = torch.empty_like(a), torch.empty_like(a)
    _lib.call("slim_fmod_probe_f32", _p(a), _p(b), a.numel(), _p(fast), _p(ref), _stream())
    return fast, ref


def floordiv_probe(a, b):
    """(the library's floor(a / b) as the scale search evaluates it, the reference's `//` written out step by step)
    element-wise for fp32 a >= 0, b > 0 (slim_floordiv_probe_f32)."""
    a, b = _chk(a, torch.float32, "a"), _chk(b, torch.float32, "b")
    fast, ref = torch.empty_like(a), torch.empty_like(a)
    _lib.call("slim_floordiv_probe_f32", _p(a), _p(b), a.numel(), _p(fast), _p(ref), _stream())
    return fast, ref


def absmax_scale(hist, num_bits):
    out = torch.empty(2, dtype=torch.float32, device=hist.device)
    _lib.call("slim_absmax_scale", _p(hist), num_bits, _p(out), _p(out[1:]), _stream())
    return out[0], out[1]


def quantize_pt(W, sf, num_bits):
    W = _chk(W, BF16, "W")
    sf = _chk(sf, torch.float32, "sf")
    codes = torch.empty(W.shape, dtype=torch.int8, device=W.device)
    _lib.call("slim_quantize_pt", _p(W), W.numel(), _p(sf), num_bits, _p(codes), _stream())
    return codes


def dequantize_pt(codes, sf, out_bf16=False):
    codes = _chk(codes, torch.int8, "codes")
    sf = _chk(sf, torch.float32, "sf")
    out = torch.empty(codes.shape, dtype=BF16 if out_bf16 else torch.float32, device=codes.device)
    _lib.call("slim_dequantize_pt", _p(codes), codes.numel(), _p(sf), _p(out), int(out_bf16), _stream())
    return out


def fake_quant_pt(W, sf, num_bits, want_codes=False):
    W = _chk(W, BF16, "W")
    sf = _chk(sf, torch.float32, "sf")
    out = torch.empty_like(W)
    codes = torch.empty(W.shape, dtype=torch.int8, device=W.device) if want_codes else None
    _lib.call("slim_fake_quant_pt", _p(W), W.numel(), _p(sf), num_bits, _p(out), _p(codes), _stream())
    return (out, codes) if want_codes else out


# ---- low-rank fit ----------------------------------------------------------------------------
def _out_or_new(out, shape, device, name):
    if out is None:
        return torch.empty(*shape, dtype=BF16, device=device)
    if tuple(out.shape) != tuple(shape) or out.dtype != BF16 or not out.is_contiguous() or not out.is_cuda:
        raise ValueError(f"{name}: `out` must be a contiguous bf16 GPU tensor of shape {tuple(shape)}")
    return out


def error_matrix(W, mask, s, sf, num_bits, want_f32=False, want_bf16=True, out=None, col_flags=None, deq_bf16=False):
    """(E fp32 or None, E bf16 or None), E = W * s - mask0((code / sf) * s) (sf None: no quantiser).
    col_flags (bool [K]): columns quantised at twice their size and halved after dequantisation (scale_important_weights).
    deq_bf16: code / sf rounded to bf16 first (the abs-max quantiser, whose scale is a bf16 tensor in the reference)."""
    W = _chk(W, BF16, "W")
    mask = _chk(mask, torch.bool, "mask")
    N, K = W.shape
    e32 = torch.empty(N, K, dtype=torch.float32, device=W.device) if want_f32 else None
    e16 = _out_or_new(out, (N, K), W.device, "error_matrix") if want_bf16 else None
    if sf is not None and (col_flags is not None or deq_bf16):
        if col_flags is not None:
            col_flags = _chk(col_flags, torch.bool, "col_flags")
        _lib.call("slim_error_matrix_cols", _p(W), _p(mask), _p(s), _p(sf), num_bits, _p(col_flags), int(bool(deq_bf16)), N, K,
                  _p(e32), _p(e16), _stream())
    else:
        _lib.call("slim_error_matrix", _p(W), _p(mask), _p(s), _p(sf), num_bits, N, K, _p(e32), _p(e16), _stream())
    return e32, e16


def important_columns(W, scaler_row, count=None):
    """Flags (bool [K], True = important) of the `count` input channels with the largest
    scaler_row * bf16(column sums of |W|); count defaults to int(0.01 * K) (slim/lora.py:132-136)."""
    W = _chk(W, BF16, "W")
    scaler_row = _chk(scaler_row, torch.float32, "scaler_row")
    N, K = W.shape
    assert scaler_row.numel() == K
    if count is None:
        count = int(0.01 * K)
    flags = torch.empty(K, dtype=torch.bool, device=W.device)
    ws = _ws(_lib.query("slim_important_columns_workspace_bytes", N, K), W.device)
    _lib.call("slim_important_columns", _p(W), _p(scaler_row), N, K, int(count), _p(flags), _p(ws), ws.numel(), _stream())
    return flags


def scale_columns(x, flags, factor, out=None):
    """bf16(x * factor) in the flagged columns of x [N, K], x elsewhere; out may be x itself."""
    x = _chk(x, BF16, "x")
    flags = _chk(flags, torch.bool, "flags")
    N, K = x.shape
    assert flags.numel() == K
    out = torch.empty_like(x) if out is None else _out_or_new(out, (N, K), x.device, "scale_columns")
    _lib.call("slim_scale_columns_bf16", _p(x), _p(flags), N, K, float(factor), _p(out), _stream())
    return out


def scale_columns_f32_(x, flags, factor):
    """x [N, K] fp32 times factor in the flagged columns, in place."""
    _chk(x, torch.float32, "x")
    if not x.is_contiguous():
        raise ValueError("scale_columns_f32_: x must be contiguous")
    flags = _chk(flags, torch.bool, "flags")
    _lib.call("slim_scale_columns_f32", _p(x), _p(flags), x.shape[0], x.shape[1], float(factor), _stream())
    return x


def error_matrix_deq(W, Wdeq, mask, s, out=None):
    """E (bf16) = W * s - mask0(Wdeq * s) for a dequantised weight given as a tensor."""
    W = _chk(W, BF16, "W")
    Wdeq = _chk(Wdeq, BF16, "Wdeq")
    mask = _chk(mask, torch.bool, "mask")
    N, K = W.shape
    e16 = _out_or_new(out, (N, K), W.device, "error_matrix_deq")
    _lib.call("slim_error_matrix_deq", _p(W), _p(Wdeq), _p(mask), _p(s), N, K, None, _p(e16), _stream())
    return e16


def transpose(x, out=None):
    x = _chk(x, BF16, "x")
    R, Cc = x.shape
    out = _out_or_new(out, (Cc, R), x.device, "transpose")
    _lib.call("slim_transpose_bf16", _p(x), R, Cc, _p(out), _stream())
    return out


def gemm_nt(A, Bt, alpha=1.0, beta=0.0, C32=None, out_f32=False, out_bf16=False, out_t=False, C16=None, mirrored=False):
    """C = alpha * A @ Bt.T (+ beta * C32). Returns (C fp32 | None, C bf16 | None, C^T bf16 | None).  C16 (optional): a
    contiguous bf16 [M, N] tensor that receives the bf16 result (implies out_bf16).  mirrored=True (Bt is A, bf16 output
    only): A A^T with the tiles on and above the diagonal computed and mirrored into the rest (C^T returned is C itself)."""
    A = _chk(A, BF16, "A")
    Bt = _chk(Bt, BF16, "Bt")
    M, Kd = A.shape
    N = Bt.shape[0]
    assert Bt.shape[1] == Kd
    dev = A.device
    if C32 is None and out_f32:
        C32 = torch.empty(M, N, dtype=torch.float32, device=dev)
    if C16 is not None:
        _chk(C16, BF16, "C16")
        if tuple(C16.shape) != (M, N) or not C16.is_contiguous():
            raise ValueError("C16 must be a contiguous [M, N] tensor")
    elif out_bf16:
        C16 = torch.empty(M, N, dtype=BF16, device=dev)
    Ct = torch.empty(N, M, dtype=BF16, device=dev) if out_t else None
    if mirrored:
        if Bt.data_ptr() != A.data_ptr() or C32 is not None or M != N:
            raise ValueError("mirrored: A A^T with a bf16 output only")
        C16 = C16 if C16 is not None else torch.empty(M, N, dtype=BF16, device=dev)
        Ct = C16
    _lib.call("slim_gemm_nt_bf16", _p(A), Kd, _p(Bt), Kd, M, N, Kd, float(alpha), float(beta), _p(C32), N, _p(C16), N,
              _p(Ct), M, _stream())
    return C32, C16, Ct


def gemm_nt_splitk(A, Bt, splits, alpha=1.0):
    """fp32 and bf16 C = alpha * A @ Bt.T with the reduction split into `splits` deterministic slices."""
    A = _chk(A, BF16, "A")
    Bt = _chk(Bt, BF16, "Bt")
    M, Kd = A.shape
    N = Bt.shape[0]
    C32 = torch.empty(M, N, dtype=torch.float32, device=A.device)
    C16 = torch.empty(M, N, dtype=BF16, device=A.device)
    slab = torch.empty(max(splits, 1) * M * N, dtype=torch.float32, device=A.device)
    _lib.call("slim_gemm_nt_bf16_splitk", _p(A), Kd, _p(Bt), Kd, M, N, Kd, float(alpha), _p(C32), N, _p(C16), N,
              splits, _p(slab), _stream())
    return C32, C16


def gemm_nt_splitk_t(A, Bt, splits, alpha=1.0):
    """(C bf16, C^T bf16) = alpha * A @ Bt.T with the reduction split into `splits` deterministic slices."""
    A = _chk(A, BF16, "A")
    Bt = _chk(Bt, BF16, "Bt")
    M, Kd = A.shape
    N = Bt.shape[0]
    C16 = torch.empty(M, N, dtype=BF16, device=A.device)
    Ct = torch.empty(N, M, dtype=BF16, device=A.device)
    slab = torch.empty(max(splits, 1) * M * N, dtype=torch.float32, device=A.device)
    _lib.call("slim_gemm_nt_bf16_splitk_t", _p(A), Kd, _p(Bt), Kd, M, N, Kd, float(alpha), _p(C16), N, _p(Ct), M, splits,
              _p(slab), _stream())
    return C16, Ct


def tridiagonalize(G):
    """(d [n], e [n-1], Q [n, n]) with Q.T @ G @ Q = tridiag(d, e) for a symmetric fp32 matrix G."""
    G = _chk(G, torch.float32, "G").clone()
    n = G.shape[0]
    if G.dim() != 2 or G.shape[1] != n:
        raise ValueError("G must be square")
    d = torch.empty(n, dtype=torch.float32, device=G.device)
    e = torch.empty(n, dtype=torch.float32, device=G.device)
    Q = torch.empty(n, n, dtype=torch.float32, device=G.device)
    ws = torch.empty(5 * n, dtype=torch.float32, device=G.device)
    _lib.call("slim_tridiagonalize_f32", _p(G), n, _p(d), _p(e), _p(Q), _p(ws), ws.numel() * 4, _stream())
    return d, e[:n - 1], Q


def eigh(G, top=None):
    """(evals ascending [n], vecs [n, n]) of the symmetric fp32 matrix G; row n-1-i of vecs is the eigenvector of the
    i-th largest eigenvalue for i < top (default: all rows), the other rows hold no result."""
    G = _chk(G, torch.float32, "G").clone()
    n = G.shape[0]
    if G.dim() != 2 or G.shape[1] != n:
        raise ValueError("G must be square")
    top = n if top is None else int(top)
    evals = torch.empty(n, dtype=torch.float32, device=G.device)
    info = torch.zeros(1, dtype=torch.int32, device=G.device)
    ws = _ws(_lib.query("slim_eigh_workspace_bytes", n), G.device)
    _lib.call("slim_eigh_f32", _p(G), n, top, _p(evals), _p(info), _p(ws), ws.numel(), _stream())
    return evals, G, info


def orthonormalize(X, Xt=None, ns_quintic=2, ns_cubic=3, cholesky_qr=False):
    """(Q [rows, l], Qt [l, rows], info): orthonormal basis of the column space of X [rows, l]."""
    X = _chk(X, BF16, "X")
    Xt = transpose(X) if Xt is None else _chk(Xt, BF16, "Xt")
    rows, l = X.shape
    Q, Qt = torch.empty_like(X), torch.empty_like(Xt)
    info = torch.empty(64, dtype=torch.int32, device=X.device)
    ws = _ws(_lib.query("slim_orthonormalize_workspace_bytes", rows, l), X.device)
    _lib.call("slim_orthonormalize_bf16", _p(X), _p(Xt), rows, l, int(bool(cholesky_qr)), ns_quintic, ns_cubic, _p(Q),
              _p(Qt), _p(info), _p(ws), ws.numel(), _stream())
    return Q, Qt, info


def lowrank_fit(E, Et, r, oversample, n_iter, seed=0, s=None, cholesky_qr=False, ns_steps=0,
                library_eigh=False, sketch_split=True):
    """Leading-r factors of E [N, K]: returns dict(U [N,r], Ut [r,N], L [K,r], S [r], info).
    E [B, N, K] (with Et [B, K, N], s [B, K]) runs B fits of identical shape in lockstep and returns
    tensors with a leading batch dimension."""
    E = _chk(E, BF16, "E")
    Et = _chk(Et, BF16, "Et")
    batched = E.dim() == 3
    B = E.shape[0] if batched else 1
    N, K = E.shape[-2:]
    if tuple(Et.shape) != ((B, K, N) if batched else (K, N)):
        raise ValueError("Et must be the transpose of E")
    l = min(r + oversample, N, K)
    l -= l % 8
    if r > l:
        raise ValueError(f"rank {r} exceeds the usable subspace dimension {l} of a {N}x{K} matrix")
    if s is not None:
        s = _chk(s, torch.float32, "s")
        if s.numel() != B * K:
            raise ValueError("s must hold one scale per input channel (and per problem)")
    dev = E.device
    lead = (B,) if batched else ()
    U = torch.empty(*lead, N, r, dtype=BF16, device=dev)
    Ut = torch.empty(*lead, r, N, dtype=BF16, device=dev)
    L = torch.empty(*lead, K, r, dtype=BF16, device=dev)
    S = torch.empty(*lead, r, dtype=torch.float32, device=dev)
    info = torch.empty(*lead, 64, dtype=torch.int32, device=dev)
    ws = _ws(_lib.query("slim_lowrank_fit_batched_workspace_bytes", B, N, K, l), dev)
    _lib.call("slim_lowrank_fit_batched", _p(E), _p(Et), B, N, K, r, l, n_iter, seed, fit_flags(cholesky_qr, library_eigh, sketch_split),
              int(ns_steps), _p(s), _p(U), _p(Ut), _p(L), _p(S), _p(info), _p(ws), ws.numel(), _stream())
    return dict(U=U, Ut=Ut, L=L, S=S, info=info, l=l)


def lora_prune_optimize(L32, R32, L_mask, T32, num_iters, lr, lr_end_factor, want_losses=False):
    """In place: num_iters Adam steps on fp32 L32 [K, r] (kept zero where L_mask) and R32 [r, N] towards T32 [K, N]."""
    L32 = _chk(L32, torch.float32, "L32")
    R32 = _chk(R32, torch.float32, "R32")
    T32 = _chk(T32, torch.float32, "T32")
    L_mask = _chk(L_mask, torch.bool, "L_mask")
    K, r = L32.shape
    N = R32.shape[1]
    if tuple(R32.shape) != (r, N) or tuple(T32.shape) != (K, N) or tuple(L_mask.shape) != (K, r):
        raise ValueError("shapes must be L [K, r], R [r, N], T [K, N], L_mask [K, r]")
    losses = torch.empty(max(num_iters, 1), dtype=torch.float32, device=L32.device) if want_losses else None
    ws = _ws(_lib.query("slim_lora_prune_workspace_bytes", K, r, N), L32.device)
    _lib.call("slim_lora_prune_optimize", _p(L32), _p(R32), _p(L_mask), _p(T32), K, r, N, int(num_iters), float(lr),
              float(lr_end_factor), _p(losses), _p(ws), ws.numel(), _stream())
    return losses[:num_iters] if want_losses else None


def fit_flags(cholesky_qr=False, library_eigh=False, sketch_split=True):
    return int(bool(cholesky_qr)) | (4 if library_eigh else 0) | (0 if sketch_split else 16)


def fit_subspace(N, K, r, oversample):
    """Subspace dimension l the fit uses for rank r (a multiple of 8 within min(N, K))."""
    l = min(r + oversample, N, K)
    return l - l % 8


def set_fit_filter(ratios, bits):
    """Request (ratios non-empty) or withdraw the Chebyshev-filtered iteration for the next lowrank_fit_front* call on this
    thread (slim_lowrank_fit_set_filter): one ratio per problem of that call."""
    import ctypes
    n = len(ratios)
    arr = (ctypes.c_float * max(n, 1))(*[float(x) for x in ratios])
    _lib.call("slim_lowrank_fit_set_filter", ctypes.addressof(arr), n, int(bits))


def lowrank_fit_front(E, Et, r, oversample, n_iter, M_out, seed=0, flags=0, ns_steps=0):
    """First half of lowrank_fit: everything up to the l x l Rayleigh-Ritz matrix, which is written to M_out (fp32 [l, l]).
    E [B, N, K] (Et [B, K, N], M_out [B, l, l]) runs B same-shape fits in lockstep.
    Returns the context lowrank_fit_back needs (it owns the workspace)."""
    import ctypes
    E = _chk(E, BF16, "E")
    Et = _chk(Et, BF16, "Et")
    M_out = _chk(M_out, torch.float32, "M_out")
    batched = E.dim() == 3
    B = E.shape[0] if batched else 1
    N, K = E.shape[-2:]
    if tuple(Et.shape) != ((B, K, N) if batched else (K, N)):
        raise ValueError("Et must be the transpose of E")
    l = fit_subspace(N, K, r, oversample)
    if r > l or tuple(M_out.shape) != ((B, l, l) if batched else (l, l)):
        raise ValueError("bad rank or Rayleigh-Ritz buffer")
    state = ctypes.c_int32(-1)
    ws = _ws(_lib.query("slim_lowrank_fit_batched_workspace_bytes", B, N, K, l), E.device)
    _lib.call("slim_lowrank_fit_front_batched", _p(E), _p(Et), B, N, K, r, l, n_iter, seed, int(flags), int(ns_steps),
              _p(M_out), ctypes.addressof(state), _p(ws), ws.numel(), _stream())
    return dict(ws=ws, state=state, N=N, K=K, r=r, l=l, n_iter=n_iter, keep=(E, Et), batch=B, batched=batched)


def tridiagonalize_batched(A):
    """A [B, n, n] symmetric fp32 (destroyed) -> d [B, n], e [B, n], Q [B, n, n] in one launch chain."""
    A = _chk(A, torch.float32, "A")
    B, n, _ = A.shape
    d = torch.empty(B, n, dtype=torch.float32, device=A.device)
    e = torch.empty(B, n, dtype=torch.float32, device=A.device)
    Q = torch.empty(B, n, n, dtype=torch.float32, device=A.device)
    ws = torch.empty(5 * n * B, dtype=torch.float32, device=A.device)
    _lib.call("slim_tridiagonalize_f32_batched", _p(A), n, B, _p(d), _p(e), _p(Q), _p(ws), ws.numel() * 4, _stream())
    return d, e, Q


def eigh_tail(d, e, Q, vecs, top):
    """Second stage of the eigen-solver for all problems of one size in one call: tridiagonal (d, e) [B, n] with the
    reductions' Q [B, n, n] (a single problem may come without the leading dimension) -> eigenvalues ascending in d, row
    n-1-i of vecs [B, n, n] = eigenvector of the i-th largest for i < top.  Returns the status words (int32 [B] on the
    device, 0 = fine)."""
    d, e, Q, vecs = (_chk(t, torch.float32, "eig") for t in (d, e, Q, vecs))
    n = d.shape[-1]
    B = d.numel() // n
    if e.numel() != B * n or Q.numel() != B * n * n or vecs.numel() != B * n * n:
        raise ValueError("eigh_tail: operands of B problems of size n x n expected")
    info = torch.zeros(B, dtype=torch.int32, device=Q.device)
    ws = _ws(_lib.query("slim_eigh_tail_workspace_bytes", n, int(top), B), Q.device)
    _lib.call("slim_eigh_tail_f32_batched", _p(d), _p(e), _p(Q), n, B, int(top), _p(vecs), _p(info), _p(ws), ws.numel(), _stream())
    return info


# rounds of the Chebyshev-filtered iteration of ritz_split (degrees of the polynomial per round)
RITZ_SPLIT_DEGREES = tuple(int(x) for x in os.environ.get("SLIM_RITZ_SPLIT_DEGREES", "8,12,20").split(","))


def ritz_split_supported(l, drop_max):
    return bool(_lib.query("slim_ritz_split_supported", int(l), int(drop_max)))


def ritz_split(M, top, keep, degrees=None, _workspace=False):
    """The fit's Rayleigh-Ritz step without an eigen-decomposition: M [B, l, l] symmetric fp32 (left intact) ->
    (vecs [B, l, l], evals [B, l], info int32 [B]) in the layout of eigh_tail: row l-1-j of vecs[b] = j-th vector of an
    orthonormal basis of the invariant subspace of the keep[b] largest eigenvalues (zero rows for keep[b] <= j < top),
    evals[b, l-1-j] its Rayleigh quotient.  keep: one int per problem."""
    import ctypes
    M = _chk(M, torch.float32, "M")
    B, l, _ = M.shape
    keep = [int(k) for k in keep]
    if len(keep) != B:
        raise ValueError("ritz_split: one `keep` per problem")
    degrees = [int(d) for d in (degrees or RITZ_SPLIT_DEGREES)]
    drop = max(l - k for k in keep)
    vecs = torch.empty_like(M)
    evals = torch.zeros(B, l, dtype=torch.float32, device=M.device)
    info = torch.empty(B, dtype=torch.int32, device=M.device)
    ws = _ws(_lib.query("slim_ritz_split_workspace_bytes", l, drop, B), M.device)
    ka = (ctypes.c_int32 * B)(*keep)
    da = (ctypes.c_int32 * len(degrees))(*degrees)
    _lib.call("slim_ritz_split_f32_batched", _p(M), l, B, int(top), ctypes.addressof(ka), ctypes.addressof(da), len(degrees), _p(vecs),
              _p(evals), _p(info), _p(ws), ws.numel(), _stream())
    if _workspace:           # (tests look at the bounds the call computed)
        return vecs, evals, info, ws
    return vecs, evals, info


def lowrank_fit_back(ctx, d, e, Q, vecs, s=None, tail_info=None):
    """Second half: d, e, Q of the tridiagonalised Rayleigh-Ritz matrix (one problem, or [B, ...] for a context of B fits)
    -> dict as lowrank_fit returns.  vecs: an [l, l] ([B, l, l]) fp32 scratch buffer (the Rayleigh-Ritz matrix's own
    slot will do).  tail_info (int32 [B]): the caller has already run eigh_tail on every problem (d and vecs hold its
    results) and passes the status words."""
    import ctypes
    N, K, r, l, B = ctx["N"], ctx["K"], ctx["r"], ctx["l"], ctx.get("batch", 1)
    dev = vecs.device
    if tail_info is not None and e is None and Q is None:
        # the eigen-stage has run elsewhere (eigh_tail or ritz_split): d = its eigenvalue array, vecs = its vectors
        d, vecs = (_chk(t, torch.float32, "eig") for t in (d, vecs))
        if vecs.numel() != B * l * l or d.numel() != B * l:
            raise ValueError("eigen-solver operands do not match the fit context")
    else:
        d, e, Q, vecs = (_chk(t, torch.float32, "eig") for t in (d, e, Q, vecs))
        if Q.numel() != B * l * l or vecs.numel() != B * l * l or d.numel() != B * l or e.numel() != B * l:
            raise ValueError("eigen-solver operands do not match the fit context")
    if tail_info is not None:
        info1 = _chk(tail_info, torch.int32, "tail_info")
        if info1.numel() != B:
            raise ValueError("tail_info must hold one status word per problem")
    else:
        info1 = eigh_tail(d.reshape(B, l), e.reshape(B, l), Q.reshape(B, l, l), vecs.reshape(B, l, l), r)
    if s is not None:
        s = _chk(s, torch.float32, "s")
        if s.numel() != B * K:
            raise ValueError("s must hold one scale per input channel (and per problem)")
    lead = (B,) if ctx.get("batched") else ()
    U = torch.empty(*lead, N, r, dtype=BF16, device=dev)
    Ut = torch.empty(*lead, r, N, dtype=BF16, device=dev)
    L = torch.empty(*lead, K, r, dtype=BF16, device=dev)
    S = torch.empty(*lead, r, dtype=torch.float32, device=dev)
    info = torch.empty(*lead, 64, dtype=torch.int32, device=dev)
    ws = ctx["ws"]
    _lib.call("slim_lowrank_fit_back_batched", B, N, K, r, l, ctx["n_iter"], ctypes.addressof(ctx["state"]), _p(vecs), _p(d),
              _p(info1), _p(s), _p(U), _p(Ut), _p(L), _p(S), _p(info), _p(ws), ws.numel(), _stream())
    return dict(U=U, Ut=Ut, L=L, S=S, info=info, l=l)


def _ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def lowrank_fit_front_group(Es, Ets, r, oversample, n_iter, M_out, seed=0, flags=0, ns_steps=0):
    """lowrank_fit_front for a lockstep group of matrices whose shapes may differ but whose short dimension agrees: Es[b]
    [N_b, K_b], Ets[b] its transpose, M_out [B, l, l].  Every member's result is that of lowrank_fit_front on it alone.
    Returns the context lowrank_fit_back_group needs."""
    import ctypes
    B = len(Es)
    Es = [_chk(E, BF16, "E") for E in Es]
    Ets = [_chk(Et, BF16, "Et") for Et in Ets]
    M_out = _chk(M_out, torch.float32, "M_out")
    shapes = [tuple(E.shape) for E in Es]
    if any(len(sh) != 2 for sh in shapes) or any(tuple(Et.shape) != (K, N) for Et, (N, K) in zip(Ets, shapes)):
        raise ValueError("Es must be matrices and Ets their transposes")
    Dm = min(shapes[0])
    if any(min(sh) != Dm for sh in shapes):
        raise ValueError("the matrices of a group must share their short dimension")
    l = fit_subspace(Dm, Dm, r, oversample)
    if r > l or l >= Dm or tuple(M_out.shape) != (B, l, l):
        raise ValueError("bad rank or Rayleigh-Ritz buffer")
    Ns = (ctypes.c_int64 * B)(*[sh[0] for sh in shapes])
    Ks = (ctypes.c_int64 * B)(*[sh[1] for sh in shapes])
    state = ctypes.c_int32(-1)
    ws = _ws(_lib.query("slim_lowrank_fit_group_workspace_bytes", ctypes.addressof(Ns), ctypes.addressof(Ks), B, l), Es[0].device)
    pe, pet = _ptr_array(Es), _ptr_array(Ets)
    _lib.call("slim_lowrank_fit_front_group", ctypes.addressof(pe), ctypes.addressof(pet), ctypes.addressof(Ns),
              ctypes.addressof(Ks), B, r, l, n_iter, seed, int(flags), int(ns_steps), _p(M_out), ctypes.addressof(state),
              _p(ws), ws.numel(), _stream())
    return dict(ws=ws, state=state, shapes=shapes, Ns=Ns, Ks=Ks, r=r, l=l, n_iter=n_iter, keep=(Es, Ets), batch=B)


def lowrank_fit_back_group(ctx, d, e, Q, vecs, ss=None, tail_info=None):
    """Second half for the group of lowrank_fit_front_group: d, e [B, l], Q, vecs [B, l, l]; ss: per-member scale vectors
    (entries may be None).  Returns one dict(U, Ut, L, S, info) per member."""
    import ctypes
    B, r, l, shapes = ctx["batch"], ctx["r"], ctx["l"], ctx["shapes"]
    dev = vecs.device
    if tail_info is not None and e is None and Q is None:
        d, vecs = (_chk(t, torch.float32, "eig") for t in (d, vecs))
        if vecs.numel() != B * l * l or d.numel() != B * l:
            raise ValueError("eigen-solver operands do not match the fit context")
    else:
        d, e, Q, vecs = (_chk(t, torch.float32, "eig") for t in (d, e, Q, vecs))
        if Q.numel() != B * l * l or vecs.numel() != B * l * l or d.numel() != B * l or e.numel() != B * l:
            raise ValueError("eigen-solver operands do not match the fit context")
    if tail_info is not None:
        info1 = _chk(tail_info, torch.int32, "tail_info")
        if info1.numel() != B:
            raise ValueError("tail_info must hold one status word per problem")
    else:
        info1 = eigh_tail(d.reshape(B, l), e.reshape(B, l), Q.reshape(B, l, l), vecs.reshape(B, l, l), r)
    ss = list(ss) if ss is not None else [None] * B
    for s, (N, K) in zip(ss, shapes):
        if s is not None and _chk(s, torch.float32, "s").numel() != K:
            raise ValueError("s must hold one scale per input channel")
    U = [torch.empty(N, r, dtype=BF16, device=dev) for N, K in shapes]
    Ut = [torch.empty(r, N, dtype=BF16, device=dev) for N, K in shapes]
    L = [torch.empty(K, r, dtype=BF16, device=dev) for N, K in shapes]
    S = torch.empty(B, r, dtype=torch.float32, device=dev)
    info = torch.empty(B, 64, dtype=torch.int32, device=dev)
    ws = ctx["ws"]
    ps, pu, put, pl = _ptr_array(ss), _ptr_array(U), _ptr_array(Ut), _ptr_array(L)
    _lib.call("slim_lowrank_fit_back_group", ctypes.addressof(ctx["Ns"]), ctypes.addressof(ctx["Ks"]), B, r, l,
              ctypes.addressof(ctx["state"]), _p(vecs), _p(d), _p(info1), ctypes.addressof(ps), ctypes.addressof(pu),
              ctypes.addressof(put), ctypes.addressof(pl), _p(S), _p(info), _p(ws), ws.numel(), _stream())
    return [dict(U=U[b], Ut=Ut[b], L=L[b], S=S[b], info=info[b], l=l) for b in range(B)]


# ---- the Gram-side fit in pieces (layers sharded over several GPUs) ---------------------------------------
def gramfit_iterate(C16, l, n_iter, seed=0, flags=0, cond_e=0):
    """Subspace iteration on the symmetric bf16 matrix C16 [Dm, Dm] -> (P [Dm, l], Pt [l, Dm], info int32 [64])."""
    C16 = _chk(C16, BF16, "C16")
    Dm = C16.shape[0]
    P = torch.empty(Dm, l, dtype=BF16, device=C16.device)
    Pt = torch.empty(l, Dm, dtype=BF16, device=C16.device)
    info = torch.empty(64, dtype=torch.int32, device=C16.device)
    ws = _ws(_lib.query("slim_gramfit_workspace_bytes", Dm, l), C16.device)
    _lib.call("slim_gramfit_iterate", _p(C16), Dm, int(l), int(n_iter), int(seed), int(flags), int(cond_e), _p(P), _p(Pt), _p(info),
              _p(ws), ws.numel(), _stream())
    return P, Pt, info


def gramfit_iterate_batched(C16, l, n_iter, seed=0, flags=0, cond_e=0):
    """gramfit_iterate for B same-size matrices in lockstep: C16 [B, Dm, Dm] -> (P [B, Dm, l], Pt [B, l, Dm], info int32
    [B, 64], scal fp32 [B, 4] for gramfit_check).  A request left by set_fit_filter on this thread (B ratios) is honoured;
    the filtered iteration shifts the diagonals of C16 in place."""
    C16 = _chk(C16, BF16, "C16")
    B, Dm, _ = C16.shape
    dev = C16.device
    P = torch.empty(B, Dm, l, dtype=BF16, device=dev)
    Pt = torch.empty(B, l, Dm, dtype=BF16, device=dev)
    info = torch.empty(B, 64, dtype=torch.int32, device=dev)
    scal = torch.empty(B, 4, dtype=torch.float32, device=dev)
    ws = _ws(_lib.query("slim_gramfit_batched_workspace_bytes", B, Dm, l), dev)
    _lib.call("slim_gramfit_iterate_batched", _p(C16), B, Dm, int(l), int(n_iter), int(seed), int(flags), int(cond_e), _p(P), _p(Pt),
              _p(info), _p(scal), _p(ws), ws.numel(), _stream())
    return P, Pt, info, scal


def gramfit_check(evals, r, cond_e, scal, info):
    """The report words of finished fits (conditioning request, measured spread, where the cut sits: the last three of info
    int32 [B, 64], written in place) from the eigenvalues evals [B, l] of their Rayleigh-Ritz matrices; scal: from
    gramfit_iterate_batched, or None."""
    evals = _chk(evals, torch.float32, "evals")
    _chk(info, torch.int32, "info")
    B, l = evals.shape
    if info.numel() != B * 64 or (scal is not None and scal.numel() != B * 4):
        raise ValueError("gramfit_check: info must hold 64 words and scal 4 floats per problem")
    _lib.call("slim_gramfit_check", _p(evals), l, int(r), B, int(cond_e), _p(scal), _p(info), _stream())
    return info


def gramfit_project(X, Pt, M32=None):
    """X [rows, Dm] (a shard of E or E^T), Pt [l, Dm] -> (Z = X P [rows, l], Zt [l, rows], M32 = Z^T Z fp32 [l, l]; written
    into the given contiguous tensor when M32 is passed)."""
    X = _chk(X, BF16, "X")
    Pt = _chk(Pt, BF16, "Pt")
    rows, Dm = X.shape
    l = Pt.shape[0]
    Z = torch.empty(rows, l, dtype=BF16, device=X.device)
    Zt = torch.empty(l, rows, dtype=BF16, device=X.device)
    if M32 is None:
        M32 = torch.empty(l, l, dtype=torch.float32, device=X.device)
    elif M32.dtype != torch.float32 or tuple(M32.shape) != (l, l) or not M32.is_contiguous():
        raise ValueError("M32 must be a contiguous fp32 [l, l] tensor")
    ws = _ws(_lib.query("slim_gramfit_workspace_bytes", rows, l), X.device)
    _lib.call("slim_gramfit_project", _p(X), rows, Dm, l, _p(Pt), _p(Z), _p(Zt), _p(M32), _p(ws), ws.numel(), _stream())
    return Z, Zt, M32


def gramfit_factors(right, P, Z, vecs, evals, r, s=None, want_dm_t=False, want_rows_t=False, info=None, cond_e=0):
    """The factors from the Rayleigh-Ritz eigenvectors; see slim_gramfit_factors.  Returns (dm_out [Dm, r], dm_out_t |
    None, rows_out [rows, r], rows_out_t | None, S [r]).  info (int32 [64], optional): its last word receives the
    conditioning check for an iteration sized for cond_e."""
    P = _chk(P, BF16, "P")
    Z = _chk(Z, BF16, "Z")
    vecs = _chk(vecs, torch.float32, "vecs")
    evals = _chk(evals, torch.float32, "evals")
    Dm, l = P.shape
    rows = Z.shape[0]
    dev = P.device
    dm_out = torch.empty(Dm, r, dtype=BF16, device=dev)
    dm_t = torch.empty(r, Dm, dtype=BF16, device=dev) if want_dm_t else None
    rows_out = torch.empty(rows, r, dtype=BF16, device=dev)
    rows_t = torch.empty(r, rows, dtype=BF16, device=dev) if want_rows_t else None
    S = torch.empty(r, dtype=torch.float32, device=dev)
    if s is not None:
        s = _chk(s, torch.float32, "s")
    ws = _ws(2 * ((l * l * 2 + 255) // 256 * 256), dev)
    _lib.call("slim_gramfit_factors", int(bool(right)), _p(P), _p(Z), Dm, rows, l, int(r), _p(vecs), _p(evals), _p(s), _p(dm_out),
              _p(dm_t), _p(rows_out), _p(rows_t), _p(S), int(cond_e), _p(info), _p(ws), ws.numel(), _stream())
    return dm_out, dm_t, rows_out, rows_t, S


def scale_rows(L, s):
    """In place: L[k, :] = bf16(L[k, :] / bf16(s[k])) (the division of lora_left by sqrt(scaler_row))."""
    L = _chk(L, BF16, "L")
    s = _chk(s, torch.float32, "s")
    _lib.call("slim_scale_rows_bf16", _p(L), _p(s), L.shape[0], L.shape[1], _stream())
    return L


def lowrank_residual(U, L, W, mask, want_lr=False, out=None):
    U = _chk(U, BF16, "U")
    L = _chk(L, BF16, "L")
    W = _chk(W, BF16, "W")
    if mask is not None:
        mask = _chk(mask, torch.bool, "mask")
    N, K = W.shape
    r = U.shape[1]
    if out is not None and (out.dtype != BF16 or tuple(out.shape) != tuple(W.shape) or not out.is_contiguous()):
        raise ValueError("out must be a contiguous bf16 tensor of W's shape")
    Wn = torch.empty_like(W) if out is None else out
    LR = torch.empty_like(W) if want_lr else None
    _lib.call("slim_lowrank_residual", _p(U), _p(L), N, K, r, _p(W), _p(mask), _p(Wn), _p(LR), _stream())
    return (Wn, LR) if want_lr else Wn


def lora_finish(U, L, W, mask, rank, quant_mode, num_bits, prescale):
    """The tail of add_lora in one call (slim_lora_finish): returns (compressed weight bf16 [N, K], scaling factor fp32 [1]
    or None, lora_left [K, rank], lora_right [rank, N]).  quant_mode: 0 none, 1 SLiM-Quant, 2 abs-max."""
    U = _chk(U, BF16, "U")
    L = _chk(L, BF16, "L")
    W = _chk(W, BF16, "W")
    mask = _chk(mask, torch.bool, "mask")
    N, K = W.shape
    r8 = U.shape[1]
    dev = W.device
    Wq = torch.empty_like(W)
    sf = torch.empty(1, dtype=torch.float32, device=dev) if quant_mode else None
    left = torch.empty(K, rank, dtype=BF16, device=dev)
    right = torch.empty(rank, N, dtype=BF16, device=dev)
    ws = _ws(_lib.query("slim_lora_finish_workspace_bytes", N, K), dev)
    _lib.call("slim_lora_finish", _p(U), _p(L), N, K, r8, int(rank), _p(W), _p(mask), int(quant_mode), int(num_bits), int(bool(prescale)),
              _p(Wq), _p(sf), _p(left), _p(right), _p(ws), ws.numel(), _stream())
    return Wq, sf, left, right


def outlier_project(Et, UOt):
    """Ct [K, tp] = bf16(Et @ UOt.T): the columns of E projected on the outlier basis (slim_outlier_project)."""
    Et, UOt = _chk(Et, BF16, "Et"), _chk(UOt, BF16, "UOt")
    K, N = Et.shape
    tp = UOt.shape[0]
    Ct = torch.empty(K, tp, dtype=BF16, device=Et.device)
    _lib.call("slim_outlier_project", _p(Et), _p(UOt), K, N, tp, _p(Ct), _stream())
    return Ct


def deflate_apply(E, UO, Ct, out=None):
    """(E2, E2t) = (bf16(E - bf16(UO @ Ct.T)), its transpose) in one pass (slim_deflate_apply)."""
    E, UO, Ct = _chk(E, BF16, "E"), _chk(UO, BF16, "UO"), _chk(Ct, BF16, "Ct")
    N, K = E.shape
    E2 = _out_or_new(out, (N, K), E.device, "deflate_apply")
    E2t = torch.empty(K, N, dtype=BF16, device=E.device)
    _lib.call("slim_deflate_apply", _p(E), _p(UO), _p(Ct), N, K, UO.shape[1], _p(E2), _p(E2t), _stream())
    return E2, E2t


def lora_assemble(UO, Ct, s, fitU, fitL, t, rank, r8):
    """(U [N, r8], Ut [r8, N], L [K, r8]) of slim_lora_assemble: the exact part of the split-off outlier columns in front of
    the randomized fit's factors."""
    UO, Ct, fitU, fitL = _chk(UO, BF16, "UO"), _chk(Ct, BF16, "Ct"), _chk(fitU, BF16, "fitU"), _chk(fitL, BF16, "fitL")
    N, tp = UO.shape
    K, rf = fitL.shape
    dev = UO.device
    U = torch.empty(N, r8, dtype=BF16, device=dev)
    Ut = torch.empty(r8, N, dtype=BF16, device=dev)
    L = torch.empty(K, r8, dtype=BF16, device=dev)
    _lib.call("slim_lora_assemble", _p(UO), _p(Ct), _p(s), _p(fitU), _p(fitL), N, K, tp, int(t), rf, int(rank), int(r8), _p(U), _p(Ut),
              _p(L), _stream())
    return U, Ut, L


def lora_adapters(U, L, rank, prescale=True):
    """(lora_left [K, rank], lora_right [rank, N]) from the factors U [N, r8], L [K, r8]: cut to `rank` columns, multiplied
    by sqrt(rank) when prescale (slim/prune.py:296-301), U transposed (slim_lora_adapters)."""
    U, L = _chk(U, BF16, "U"), _chk(L, BF16, "L")
    N, r8 = U.shape
    K = L.shape[0]
    if L.shape[1] != r8:
        raise ValueError("U and L must have the same number of columns")
    left = torch.empty(K, rank, dtype=BF16, device=U.device)
    right = torch.empty(rank, N, dtype=BF16, device=U.device)
    _lib.call("slim_lora_adapters", _p(U), _p(L), N, K, r8, int(rank), int(bool(prescale)), _p(left), _p(right), _stream())
    return left, right


def outlier_basis(Et, cols, t):
    """(UO [N, tp], UOt [tp, N]): an orthonormal basis of the columns cols[:t] of E, read as rows of Et [K, N] (bf16); cols int32,
    padded to tp entries (a multiple of 8, at most 64); the columns >= t of the basis are zero."""
    Et = _chk(Et, BF16, "Et")
    cols = _chk(cols, torch.int32, "cols")
    N, tp = Et.shape[1], cols.numel()
    UO = torch.empty(N, tp, dtype=BF16, device=Et.device)
    UOt = torch.empty(tp, N, dtype=BF16, device=Et.device)
    ws = _ws(_lib.query("slim_outlier_basis_workspace_bytes", tp), Et.device)
    _lib.call("slim_outlier_basis", _p(Et), N, _p(cols), int(t), tp, _p(UO), _p(UOt), _p(ws), ws.numel(), _stream())
    return UO, UOt


def lora_prepare(W, scaler_row, prune_n, prune_m, mask, quant_mode, num_bits, scale_error, outliers=None):
    """The head of add_lora in one call (slim_lora_prepare).  mask: a bool [N, K] tensor to use, or None to compute the N:M
    mask from |W| * sqrt(scaler_row).  outliers: (int32 indices padded to a multiple of 8, number t of real ones) or None.
    Returns dict(mask, s, sf, E, Et, deflated) with deflated = (E2, E2t, UO, Ct, info words) or None."""
    W = _chk(W, BF16, "W")
    scaler_row = _chk(scaler_row, torch.float32, "scaler_row")
    N, K = W.shape
    dev = W.device
    given = mask is not None
    mask = _chk(mask, torch.bool, "mask") if given else torch.empty(N, K, dtype=torch.bool, device=dev)
    s = torch.empty(K, dtype=torch.float32, device=dev)
    sf = torch.empty(1, dtype=torch.float32, device=dev) if quant_mode else None
    E = torch.empty(N, K, dtype=BF16, device=dev)
    Et = torch.empty(K, N, dtype=BF16, device=dev)
    t, tp, idx = 0, 0, None
    UO = UOt = Ct = E2 = E2t = info = None
    if outliers is not None and outliers[1] > 0:
        idx, t = outliers
        idx = _chk(idx, torch.int32, "outlier indices")
        tp = idx.numel()
        UO = torch.empty(N, tp, dtype=BF16, device=dev)
        UOt = torch.empty(tp, N, dtype=BF16, device=dev)
        Ct = torch.empty(K, tp, dtype=BF16, device=dev)
        E2 = torch.empty(N, K, dtype=BF16, device=dev)
        E2t = torch.empty(K, N, dtype=BF16, device=dev)
        info = torch.zeros(64, dtype=torch.int32, device=dev)
    ws = _ws(_lib.query("slim_lora_prepare_workspace_bytes", N, K, tp), dev)
    _lib.call("slim_lora_prepare", _p(W), _p(scaler_row), N, K, 0 if given else int(prune_n), int(prune_m), _p(mask), int(quant_mode),
              int(num_bits), int(bool(scale_error)), _p(s), _p(sf), _p(E), _p(Et), _p(idx), int(t), int(tp), _p(UO), _p(UOt), _p(Ct),
              _p(E2), _p(E2t), _p(ws), ws.numel(), _stream())
    return dict(mask=mask, s=s, sf=sf, E=E, Et=Et, deflated=(E2, E2t, UO, Ct, info) if t else None, UOt=UOt)


def residual_weight(W, LR, mask, want_hist=False):
    W = _chk(W, BF16, "W")
    LR = _chk(LR, BF16, "LR")
    mask = _chk(mask, torch.bool, "mask")
    Wn = torch.empty_like(W)
    hist = torch.empty(32768, dtype=torch.int32, device=W.device) if want_hist else None
    _lib.call("slim_residual_weight", _p(W), _p(LR), _p(mask), W.numel(), _p(Wn), _p(hist), _stream())
    return (Wn, hist) if want_hist else Wn


def randn_bf16(shape, seed, device):
    out = torch.empty(shape, dtype=BF16, device=device)
    _lib.call("slim_randn_bf16", _p(out), out.numel(), seed, _stream())
    return out


# ---- group quantisation ------------------------------------------------------------------------
def group_absmax(x, block):
    x = _chk(x, BF16, "x")
    M, N = x.shape
    assert M % block == 0 and N % block == 0, "Input matrix size is not divisible by block size"
    al = torch.empty(M // block, N, dtype=BF16, device=x.device)
    _lib.call("slim_group_absmax", _p(x), M, N, block, _p(al), _stream())
    return al


def group_quantize(x, alphas, bits):
    x = _chk(x, BF16, "x")
    alphas = _chk(alphas, BF16, "alphas")
    M, N = x.shape
    block = M // alphas.shape[0]
    codes = torch.empty(M, N, dtype=torch.int8, device=x.device)
    _lib.call("slim_group_quantize", _p(x), _p(alphas), M, N, block, bits, _p(codes), _stream())
    return codes


def group_dequantize(codes, alphas, bits):
    codes = _chk(codes, torch.int8, "codes")
    alphas = _chk(alphas, BF16, "alphas")
    M, N = codes.shape
    block = M // alphas.shape[0]
    out = torch.empty(M, N, dtype=BF16, device=codes.device)
    _lib.call("slim_group_dequantize", _p(codes), _p(alphas), M, N, block, bits, _p(out), _stream())
    return out


def group_fake_quant(x, block, bits, want_alphas=False, want_codes=False):
    """deq(quant(x)) with one scale per (block rows x 1 column); optionally also the scales bf16 [M / block, N]
    and the int8 codes [M, N].  Returns out | (out, alphas) | (out, alphas, codes)."""
    x = _chk(x, BF16, "x")
    M, N = x.shape
    assert M % block == 0 and N % block == 0, "Input matrix size is not divisible by block size"
    out = torch.empty_like(x)
    al = torch.empty(M // block, N, dtype=BF16, device=x.device) if (want_alphas or want_codes) else None
    codes = torch.empty(M, N, dtype=torch.int8, device=x.device) if want_codes else None
    _lib.call("slim_group_fake_quant", _p(x), M, N, block, bits, _p(out), _p(al), _p(codes), _stream())
    if want_codes:
        return out, al, codes
    return (out, al) if want_alphas else out


# ---- activation fake-quantisation --------------------------------------------------------------
def input_fake_quant_fp8(x):
    x = _chk(x, BF16, "x")
    out = torch.empty_like(x)
    scratch = torch.empty(1, dtype=torch.int32, device=x.device)
    _lib.call("slim_input_fake_quant_fp8", _p(x), x.numel(), _p(out), _p(scratch), _stream())
    return out


def input_fake_quant_int(x, num_bits, group_size, eps=1e-4):
    x = _chk(x, BF16, "x")
    group = x.shape[-1] if group_size == -1 else group_size
    assert x.numel() % group == 0
    out = torch.empty_like(x)
    _lib.call("slim_input_fake_quant_int", _p(x), x.numel() // group, group, num_bits, float(eps), _p(out), _stream())
    return out


# ---- packed 2:4 int4 weight and the forward pass from it ----------------------------------------------

def packed24_bytes(N, K):
    import ctypes
    v, m = ctypes.c_int64(0), ctypes.c_int64(0)
    _lib.call("slim_packed24_bytes", N, K, ctypes.addressof(v), ctypes.addressof(m))
    return v.value, m.value


def pack24(W, sf, check=True):
    """(vals uint8, meta uint8) packed form of a 2:4-sparse 4-bit weight W = bf16(c / sf).  Synchronises once to
    read the status word and raises ValueError when W is not of that form; check=False returns (vals, meta, status)
    without synchronising (status: int32 device tensor, see pack24_raise)."""
    W = _chk(W, BF16, "W")
    sf = _chk(sf, torch.float32, "sf").reshape(-1)[:1]
    N, K = W.shape
    vb, mb = packed24_bytes(N, K)
    vals = torch.empty(vb, dtype=torch.uint8, device=W.device)
    meta = torch.empty(mb, dtype=torch.uint8, device=W.device)
    status = torch.empty(1, dtype=torch.int32, device=W.device)
    _lib.call("slim_pack24_int4", _p(W), _p(sf), N, K, _p(vals), _p(meta), _p(status), _stream())
    if not check:
        return vals, meta, status
    pack24_raise(int(status.item()))
    return vals, meta


def pack24_raise(st):
    """Raise for a non-zero status word of pack24."""
    if st & 1:
        raise ValueError("pack24: the weight is not 2:4 sparse (a group of 4 input channels holds more than two non-zeros)")
    if st & 2:
        raise ValueError("pack24: the weight holds values that are not 4-bit quantisation levels of the given scaling factor")


def unpack24(vals, meta, sf, N, K):
    vals = _chk(vals, torch.uint8, "vals")
    meta = _chk(meta, torch.uint8, "meta")
    sf = _chk(sf, torch.float32, "sf").reshape(-1)[:1]
    W = torch.empty(N, K, dtype=BF16, device=vals.device)
    _lib.call("slim_unpack24_int4", _p(vals), _p(meta), _p(sf), N, K, _p(W), _stream())
    return W


def linear24_forward(x, vals, meta, sf, N, bias=None, LeffT=None, ReffT=None, r=0, a_div=1.0, ws=None):
    """y [M, N] = compressed-Linear forward of x [M, K] (bf16) from the packed weight (+ adapters)."""
    x = _chk(x, BF16, "x")
    M, K = x.shape
    vals = _chk(vals, torch.uint8, "vals")
    meta = _chk(meta, torch.uint8, "meta")
    sf = _chk(sf, torch.float32, "sf")
    if vals.numel() * 4 != N * K or meta.numel() * 8 != N * K:
        raise ValueError("packed weight does not match N x K")
    if bias is not None:
        bias = _chk(bias, BF16, "bias")
    if r:
        LeffT = _chk(LeffT, BF16, "LeffT")
        ReffT = _chk(ReffT, BF16, "ReffT")
        r8, r128 = -(-r // 8) * 8, -(-r // 128) * 128
        if tuple(LeffT.shape) != (r128, K) or tuple(ReffT.shape) != (N, r8):
            raise ValueError(f"adapter operands must be [{r128}, {K}] and [{N}, {r8}]")
    need = _lib.query("slim_linear24_workspace_bytes", M, N, K, r)
    if ws is None or ws.numel() < need:
        ws = _ws(need, x.device)
    y = torch.empty(M, N, dtype=BF16, device=x.device)
    _lib.call("slim_linear24_forward", _p(x), K, M, K, _p(vals), _p(meta), _p(sf), N, _p(bias), _p(LeffT), _p(ReffT), r,
              float(a_div), _p(y), N, _p(ws), ws.numel(), _stream())
    return y


def pack_q4_operand(codes_t, alphas_t, rows_pad=None):
    """Kernel layout of a 4-bit group-quantised operand: codes_t int8 [rows, Kd] (reduction dimension last) ->
    uint8 [rows_pad, Kd / 2] with the even channel in the low nibble; alphas_t bf16 [rows, Kd / block] ->
    [rows_pad, Kd / block].  Rows beyond `rows` are zero."""
    rows, Kd = codes_t.shape
    rows_pad = rows if rows_pad is None else rows_pad
    c = codes_t.to(torch.int16) & 15
    packed = (c[:, 0::2] | (c[:, 1::2] << 4)).to(torch.uint8)
    qc = torch.zeros(rows_pad, Kd // 2, dtype=torch.uint8, device=codes_t.device)
    qc[:rows] = packed
    qa = torch.zeros(rows_pad, alphas_t.shape[1], dtype=BF16, device=codes_t.device)
    qa[:rows] = alphas_t
    return qc.contiguous(), qa.contiguous()


def linear24_forward_q4(x, vals, meta, sf, N, bias, qL, qR, r, qblock, a_div, ws=None):
    """Small-batch (M <= 16) forward with 4-bit group-quantised adapters; qL / qR = (codes, alphas) from pack_q4_operand."""
    x = _chk(x, BF16, "x")
    M, K = x.shape
    if bias is not None:
        bias = _chk(bias, BF16, "bias")
    need = _lib.query("slim_linear24_workspace_bytes", M, N, K, r)
    if ws is None or ws.numel() < need:
        ws = _ws(need, x.device)
    y = torch.empty(M, N, dtype=BF16, device=x.device)
    _lib.call("slim_linear24_forward_q4", _p(x), K, M, K, _p(vals), _p(meta), _p(sf), N, _p(bias), _p(qL[0]), _p(qL[1]),
              _p(qR[0]), _p(qR[1]), r, int(qblock), float(a_div), _p(y), N, _p(ws), ws.numel(), _stream())
    return y
