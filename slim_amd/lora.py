"""Saliency-weighted low-rank adapters (mirror of reference slim/lora.py: add_lora :105-213,
quantize_lora :56-101), with the full SVD replaced by a randomized rank-r fit on MFMA."""
import os

import torch
import tqdm.auto as tqdm

from . import ops
from .quantization.quantization import Quantizer as AutoQuantizer
from .utils import get_layers_list, find_layers

# Randomized-fit parameters: subspace dimension l = rank + max(FIT_MIN_OVERSAMPLE, rank // FIT_OVERSAMPLE_DIV), and the
# number of subspace iterations.  SLiM error matrices have a flat spectrum (sigma_r / sigma_(r+128) = 1.03 - 1.06 on
# LLaMA-2-7B / 13B-shaped layers), so the excess over the truncated SVD only halves per iteration: measured on
# 4096 x 11008 with rank 512 (tests/test_fullsize_gpu.py, tools/fit_accuracy.py) 2e-3 after 6 iterations, 3.4e-4 after
# 10 with 64 extra columns and -6e-5 with 128, against a budget of 1e-3.  64 extra columns (rank // 8 at rank 512) are
# taken: every product, orthonormalisation and the l-launch chain of the eigen-solver scale with l (one fit 10.4 -> 9.2 ms).
FIT_OVERSAMPLE_DIV = 8
FIT_MIN_OVERSAMPLE = 64
FIT_ITERS = 10
# larger subspaces (rank 1024 on the 8192 x 28672 layers of a 70B-shaped model: l = 1152) meet flatter spectra: measured
# excess there 9.8e-4 after 10 iterations, 6.5e-4 after 12, 4.5e-4 after 14 (0.7 ms of a 24 ms fit each)
FIT_ITERS_LARGE_EXTRA = 4
FIT_LARGE_L = 768


def fit_iters(l):
    """Subspace iterations for a fit with subspace dimension l."""
    return FIT_ITERS + (FIT_ITERS_LARGE_EXTRA if l > FIT_LARGE_L else 0)

FIT_SEED = 0
# orthonormalise the iterates by Cholesky-QR through rocSOLVER instead of Newton-Schulz on MFMA
FIT_CHOLESKY_QR = os.environ.get("SLIM_FIT_CHOLESKY_QR", "0") == "1"


# Rayleigh-Ritz eigenproblem through rocSOLVER syevd instead of the library's own one-launch-per-column solver
FIT_LIBRARY_EIGH = os.environ.get("SLIM_FIT_LIBRARY_EIGH", "0") == "1"
# split-K for the sketch products whose tile grid under-fills the chip
FIT_SKETCH_SPLIT = os.environ.get("SLIM_FIT_SKETCH_SPLIT", "1") == "1"


def fit_ns_steps(spread, hint=0, floor=0):
    """The `ns_steps` argument of slim_lowrank_fit.  Bits 0-7: log2 of the conditioning the fit's Newton-Schulz
    orthonormalisations are sized for, from the spread max(s) / median(s) of the saliency scales that are left after the outlier columns
    have been split off.  The error matrix is (benign matrix) x diag(s): its largest singular values come from the columns
    with the largest scale, the bulk from the typical ones, so the conditioning of an iterate inside the leading subspace
    is of the order of the spread (measured: about a third of it, so half the spread is assumed).  The fit checks the
    assumption against the spectrum it finds (status word SLIM_FIT_INFO_COND_WORD) and the driver repeats a fit whose
    estimate was too low with `floor`, a lower bound of the level, set to what the fit reported (prune.compress_linears).
    Bits 8-15: `hint`, the level a similar fit measured (level_hint_from_status), or 0; ignored when a floor is given (a
    repeated fit runs on the estimate alone)."""
    import math
    if os.environ.get("SLIM_FIT_COND_E"):          # tuning aid: fixed value
        return int(os.environ["SLIM_FIT_COND_E"])
    level = int(min(6, max(1, int(floor), math.ceil(math.log2(max(float(spread) / 2.0, 1.0))))))
    hint = hint_level(hint)
    if hint and not floor and FIT_LEVEL_HINTS:
        level |= int(min(12, max(2, hint))) << 8
    return level


# SLIM_FIT_LEVEL_HINTS=0: the driver does not size a block's fits from what the same linears of the previous block measured
FIT_LEVEL_HINTS = os.environ.get("SLIM_FIT_LEVEL_HINTS", "1") == "1"


def level_hint_from_status(word, ratio_word=0):
    """The hint for the next similar fit from the status words SLIM_FIT_INFO_LEVEL_WORD (and SLIM_FIT_INFO_RATIO_WORD) of a
    finished one (0: none), packed into one int: bits 0-7 the measured log2 spread of the Gram matrix's eigenvalues inside
    the subspace plus one level (the subspaces of the iterations before the last are not converged and a little worse
    conditioned), bits 8-15 the level word itself (4 log2 of that spread), bits 16-31 the ratio word (2048 x where the
    dropped part of the spectrum sits relative to the mean eigenvalue of the Gram matrix: what the Chebyshev-filtered
    iteration of the next similar fit is given, see filter_plan)."""
    import math
    if word <= 0:
        return 0
    return int(math.ceil(word / 4.0 + 1.0)) | (min(int(word), 255) << 8) | (min(max(int(ratio_word), 0), 65535) << 16)


def hint_level(hint):
    return int(hint) & 0xff


# Chebyshev-filtered subspace iteration (ops.set_fit_filter) for fits that carry the ratio of a similar finished fit:
# FIT_FILTER_ROUNDS rounds of [degree-2 polynomial of the Gram matrix, one orthonormalisation] instead of FIT_ITERS + 1
# rounds of [Gram matrix, orthonormalisation].  The damped interval ends at FIT_FILTER_MARGIN x the largest eigenvalue the
# previous fit dropped (relative to the mean eigenvalue of the Gram matrix); a fit whose own cut lies lower asks to be
# repeated (status word 63) and is then run without the filter.  SLIM_FIT_FILTER=0: never.
FIT_FILTER = os.environ.get("SLIM_FIT_FILTER", "1") == "1"
FIT_FILTER_ROUNDS = int(os.environ.get("SLIM_FIT_FILTER_ROUNDS", "3"))
FIT_FILTER_MARGIN = float(os.environ.get("SLIM_FIT_FILTER_MARGIN", "0.8"))
FIT_FILTER_BITS_STEP = int(os.environ.get("SLIM_FIT_FILTER_BITS_STEP", "2"))
FIT_FILTER_BITS_MIN = int(os.environ.get("SLIM_FIT_FILTER_BITS_MIN", "10"))


def filter_plan(hints, floor=0):
    """(ratios per member, bits, n_iter) of the filtered iteration for fits with these packed hints, or None when any of
    them carries no ratio (or the filter is off, or the fits are being repeated: floor > 0)."""
    import math
    if not FIT_FILTER or floor or FIT_CHOLESKY_QR or not FIT_LEVEL_HINTS:
        return None
    hints = [int(h) for h in hints]
    if not hints or any((h >> 16) <= 0 for h in hints):
        return None
    ratios = [(h >> 16) / 2048.0 * FIT_FILTER_MARGIN for h in hints]
    # largest eigenvalue over the interval's end: the measured spread (word / 4 = log2), 5 % slack, the margin
    kappa = max(2.0 ** (((h >> 8) & 0xff) / 4.0) for h in hints) * 1.05 / FIT_FILTER_MARGIN      # (the cut lies above `lo`)
    bits = int(math.ceil(math.log2(2.0 * (2.0 * kappa - 1.0) ** 2 - 1.0) + 0.5))
    bits = -(-max(bits, FIT_FILTER_BITS_MIN) // FIT_FILTER_BITS_STEP) * FIT_FILTER_BITS_STEP      # (classes: fits in lockstep must agree on it, see planned_fit_group)
    return ratios, max(2, min(bits, 12)), 2 * FIT_FILTER_ROUNDS - 1


# Columns of the error matrix whose saliency scale exceeds OUTLIER_RATIO x the median are split off and
# represented exactly before the randomized fit (at most MAX_OUTLIER_COLS of them): every factor 4 of spread that is
# left costs each of the fit's ~20 orthonormalisations one more Newton-Schulz step (three tall products), while splitting
# a handful of columns off costs one pass over the matrix.
# Splitting a column off spends one of the adapter's directions on that column alone, which is the best use of it only
# while such columns are few next to the rank: the low ratio applies while they number at most rank / 16 (on a
# 256 x 512 layer of rank 25, four split-off columns cost 2 % of reconstruction error), the high one always.
OUTLIER_RATIO = float(os.environ.get("SLIM_OUTLIER_RATIO", "12"))
OUTLIER_RATIO_ALWAYS = 64.0
MAX_OUTLIER_COLS = 64


def _round_up8(n):
    return -(-int(n) // 8) * 8


def deflate_outliers(E, Et, outliers, out=None):
    """Split the outlier columns of E [N, K] off: with Q_O an orthonormal basis of their span, E = Q_O (Q_O^T E) + E'.
    outliers = (column indices padded to a multiple of 8, number t of real outliers).  Returns (E' [N, K] (written to
    `out` when given), Q_O [N, tp], C^T = (Q_O^T E)^T [K, tp], status words of the orthonormalisation); the columns >= t
    of Q_O are zero.  The outlier columns of E' are not zeroed: they hold what bf16 rounding of Q_O and C left of those
    columns, which the fit can still pick up."""
    cols, t = outliers
    # the basis from the Cholesky factor of the normalised columns' Gram matrix (ops.outlier_basis); padding columns zero
    UO, UOt = ops.outlier_basis(Et, cols, t)
    info_o = torch.zeros(64, dtype=torch.int32, device=E.device)
    Ct = ops.outlier_project(Et, UOt)
    E2, E2t = ops.deflate_apply(E, UO, Ct, out=out)
    return E2, UO, Ct, info_o, E2t


def _assemble(N, K, rank, t, s, fitU, fitUt, fitL, fit_info, UO=None, Ct=None, info_o=None):
    """(U [N, r8], Ut [r8, N], L [K, r8], info) from the factors of the randomized fit (rank - t columns wanted) and the
    exact part of the deflated columns: r8 = rank rounded up to a multiple of 8, columns >= rank zero."""
    r8 = _round_up8(rank)
    if t == 0:
        if r8 != rank:
            fitU[:, rank:] = 0
            fitUt[rank:] = 0
            fitL[:, rank:] = 0
        return fitU, fitUt, fitL, fit_info
    # the status words of the outlier basis are all zero by construction (a vanishing pivot drops its column): the fit's own
    # words are the layer's
    U, Ut, L = ops.lora_assemble(UO, Ct, s, fitU, fitL, t, rank, r8)
    return U, Ut, L, fit_info


# The Rayleigh-Ritz step of a fit as a search for the small subspace to drop (ops.ritz_split) instead of an
# eigen-decomposition of the l x l matrix (a chain of l dependent launches); SLIM_FIT_RITZ_SPLIT=0: the eigen-solver.
# A fit that is being repeated (floor > 0: its spectrum was wider than expected) takes the eigen-solver.
FIT_RITZ_SPLIT = os.environ.get("SLIM_FIT_RITZ_SPLIT", "1") == "1"


def filter_levels(bits):
    """The `ns_steps` argument of a filtered fit of class `bits` (filter_plan): its orthonormalisations are sized by the
    class alone, so the word only tells the fit's own check what the class covers -- bits 8-15: the largest log2 spread
    of eigenvalues inside the subspace the class was sized for (a fit that finds more than that asks to be repeated);
    bits 0-7: the same, capped (unused by a filtered fit).  A function of the class only: the members of a lockstep
    chain share it whatever their own estimates say."""
    import math
    kappa = (math.sqrt((2.0 ** (bits - 0.5) + 1.0) / 2.0) + 1.0) / 2.0         # inverts bits = log2(2 (2 kappa - 1)^2 - 1) + 0.5
    # (the check asks for a repeat above covered + 0.5: half a level to a level and a half of tolerance, which the margins of
    # the Newton-Schulz schedules and the extra levels of a fit's last orthonormalisation absorb)
    covered = int(max(1, min(12, math.ceil(math.log2(max(kappa * FIT_FILTER_MARGIN / 1.05, 2.0))))))
    return min(6, covered) | (covered << 8)


def ritz_split_applies(l, drop, floor=0):
    """Whether the Rayleigh-Ritz step of a fit with subspace dimension l that drops `drop` directions goes through
    ops.ritz_split.  Only while the dropped subspace is a small part of the whole (at most a quarter: l = 576, 64 to 80
    dropped on 7B / 13B-shaped layers; l = 320, 64 dropped at OPT-125m's): it is found by block iteration without
    oversampling, which loses accuracy when most of the subspace is to be dropped (measured on 256 x 512 layers, 71 of 96
    dropped: total error +0.8 %), and a small l makes the eigen-solver's chain short anyway."""
    return bool(FIT_RITZ_SPLIT and not floor and 4 * drop <= l and ops.ritz_split_supported(l, drop))


def rayleigh_ritz(Mg, top, keeps, floor=0):
    """The eigen-stage between the two halves of a fit for B problems: Mg [B, l, l] (destroyed by the eigen-solver) ->
    (evals [B, l], vecs [B, l, l], info int32 [B]) as lowrank_fit_back / _group take them with e = Q = None.  keeps: per
    problem the number of directions the caller will use (rank minus split-off columns).  floor > 0 (a repeated fit): the
    eigen-solver."""
    B, l, _ = Mg.shape
    keeps = [min(int(k), int(top)) for k in keeps]
    if ritz_split_applies(l, max(l - k for k in keeps), floor):
        evals, vecs, info = (lambda v, e, i: (e, v, i))(*ops.ritz_split(Mg, top, keeps))
        return evals, vecs, info
    d, e, Q = ops.tridiagonalize_batched(Mg)
    info = ops.eigh_tail(d, e, Q, Mg, top)
    return d, Mg, info


def fit_error_matrix(E, Et, rank, s, outliers=None, spread=1.0, hint=0, floor=0):
    """Rank-`rank` factors of the error matrix E [N, K] (Et its transpose): U [N, r8], Ut [r8, N], L [K, r8]
    with r8 = rank rounded up to a multiple of 8 and the columns >= rank zero (the MFMA products that consume
    the factors need a reduction length that is a multiple of 8; callers slice [:, :rank] for the adapters).

    E = (benign matrix) x diag(s).  A few channels with very large activations make a few columns of E
    orders of magnitude larger than the rest; bf16 iterates cannot carry both scales, so those columns
    (`outliers` = (index tensor padded to a multiple of 8, number t of real outliers)) are deflated
    (deflate_outliers): the exact part is kept (rank t) and the randomized fit runs on the well-conditioned remainder
    with rank - t columns.  The fit returns its columns by decreasing singular value, so asking it for a multiple of 8
    and dropping the tail is the optimal truncation.  L is returned already divided by bf16(s)."""
    N, K = E.shape
    rf, over, l, t = fit_plan(N, K, rank, outliers)
    if t == 0 and rf != _round_up8(rank):           # rank above the largest multiple of 8 within min(N, K)
        raise NotImplementedError(f"rank {rank} of a {N}x{K} layer is not reachable with 8-aligned subspaces")
    UO = Ct = info_o = None
    if t:
        E, UO, Ct, info_o, Et = deflate_outliers(E, Et, outliers)
    if not (FIT_CHOLESKY_QR or FIT_LIBRARY_EIGH) and l < min(N, K):
        # front half (filtered when the hint carries a ratio), the Rayleigh-Ritz step (rayleigh_ritz), back half
        M = torch.empty(1, l, l, dtype=torch.float32, device=E.device)
        plan = filter_plan([hint], floor)
        ops.set_fit_filter(*(plan[:2] if plan else ([], 0)))
        ctx = ops.lowrank_fit_front(E, Et, rf, over, plan[2] if plan else fit_iters(l), M[0],
                                    **_fit_kwargs(spread, True, hint, floor, plan))
        evals, vecs, info = rayleigh_ritz(M, rf, [rank - t], floor)
        fit = ops.lowrank_fit_back(ctx, evals, None, None, vecs, s=s, tail_info=info)
    else:
        fit = ops.lowrank_fit(E, Et, rf, over, fit_iters(l), seed=FIT_SEED, s=s, cholesky_qr=FIT_CHOLESKY_QR,
                              ns_steps=fit_ns_steps(spread, hint, floor), library_eigh=FIT_LIBRARY_EIGH,
                              sketch_split=FIT_SKETCH_SPLIT)
    return _assemble(N, K, rank, t, s, fit["U"], fit["Ut"], fit["L"], fit["info"], UO, Ct, info_o)


def _fit_kwargs(spread, sketch_split=True, hint=0, floor=0, plan=None):
    return dict(seed=FIT_SEED, flags=ops.fit_flags(FIT_CHOLESKY_QR, FIT_LIBRARY_EIGH,
                                                    FIT_SKETCH_SPLIT and sketch_split),
                ns_steps=filter_levels(plan[1]) if plan else fit_ns_steps(spread, hint, floor))


def fit_plan(N, K, rank, outliers=None):
    """(rank the randomized fit is asked for, oversampling, subspace dimension l, number t of deflated outlier columns)
    of fit_error_matrix for these arguments; l is the size of the fit's eigenproblem."""
    r8 = _round_up8(rank)
    t = outliers[1] if outliers is not None else 0
    if t == 0 or rank - t < 8:
        rf, t = min(r8, min(N, K) - min(N, K) % 8), 0
    else:
        rf = _round_up8(rank - t)
    over = max(FIT_MIN_OVERSAMPLE, rf // FIT_OVERSAMPLE_DIV)
    return rf, over, ops.fit_subspace(N, K, rf, over), t


def fit_error_matrix_front(E, Et, rank, outliers, spread, M_out, sketch_split=True, hint=0, deflated=None, floor=0):
    """fit_error_matrix up to the Rayleigh-Ritz matrix of its randomized fit (written to M_out [l, l], l from fit_plan), so
    that the eigenproblems of several layers can share one batched tridiagonalisation.  Returns the state that
    fit_error_matrix_back completes."""
    N, K = E.shape
    rf, over, l, t = fit_plan(N, K, rank, outliers)
    if t == 0 and rf != _round_up8(rank):
        raise NotImplementedError(f"rank {rank} of a {N}x{K} layer is not reachable with 8-aligned subspaces")
    p = dict(N=N, K=K, rank=rank, t=t, UO=None, Ct=None, info_o=None)
    if t:
        E, Et, p["UO"], p["Ct"], p["info_o"] = deflated if deflated is not None else deflated_pair(E, Et, outliers)
    plan = filter_plan([hint], floor) if l < min(N, K) else None
    ops.set_fit_filter(*(plan[:2] if plan else ([], 0)))
    p["ctx"] = ops.lowrank_fit_front(E, Et, rf, over, plan[2] if plan else fit_iters(l), M_out,
                                     **_fit_kwargs(spread, sketch_split, hint, floor, plan))
    return p


def deflated_pair(E, Et, outliers):
    """(E', E'^T, Q_O, C^T, status words) of deflate_outliers: what the fit of a layer with outlier columns starts from."""
    E2, UO, Ct, info_o, E2t = deflate_outliers(E, Et, outliers)
    return E2, E2t, UO, Ct, info_o


def fit_error_matrix_front_group(Es, Ets, ranks, outliers, spread, M_out, sketch_split=True, hint=0, deflated=None, hints=None,
                                 floor=0):
    """fit_error_matrix_front for B layers in lockstep.  Their shapes, ranks and deflation sets may differ as long as the
    short dimension min(N, K) and the fit's plan (rank asked of the randomized fit, oversampling, subspace dimension, see
    fit_plan) agree: the subspace iteration then runs as one batched chain.  M_out [B, l, l].  Every member's result
    equals that of fit_error_matrix_front alone.  deflated (optional): per layer the result of deflated_pair when the caller
    has split the outlier columns off already, else None."""
    B = len(Es)
    plans = [fit_plan(E.shape[0], E.shape[1], rank, o) for E, rank, o in zip(Es, ranks, outliers)]
    rf, over, l = plans[0][:3]
    if any(pl[:3] != (rf, over, l) for pl in plans) or any(min(E.shape) != min(Es[0].shape) for E in Es):
        raise ValueError("the layers of a group must share their short dimension and the plan of their fit")
    p = dict(B=B, members=[])
    Eds, Etds = [], []
    for b in range(B):
        N, K = Es[b].shape
        t = plans[b][3]
        if t == 0 and rf != _round_up8(ranks[b]):
            raise NotImplementedError(f"rank {ranks[b]} of a {N}x{K} layer is not reachable with 8-aligned subspaces")
        m = dict(N=N, K=K, rank=ranks[b], t=t, UO=None, Ct=None, info_o=None)
        E, Et = Es[b], Ets[b]
        if t:
            have = deflated[b] if deflated is not None else None
            E, Et, m["UO"], m["Ct"], m["info_o"] = have if have is not None else deflated_pair(E, Et, outliers[b])
        Eds.append(E)
        Etds.append(Et)
        p["members"].append(m)
    plan = filter_plan(hints if hints is not None else [hint] * B, floor)
    ops.set_fit_filter(*(plan[:2] if plan else ([], 0)))
    p["ctx"] = ops.lowrank_fit_front_group(Eds, Etds, rf, over, plan[2] if plan else fit_iters(l), M_out,
                                           **_fit_kwargs(spread, sketch_split, hint, floor, plan))
    return p


def fit_error_matrix_back_group(p, d, e, Q, vecs, ss, tail_info=None):
    """Second half for a group: d, e [B, l], Q, vecs [B, l, l]; ss = per-layer scale vectors (or None).  Returns one
    (U, Ut, L, info) per layer, as fit_error_matrix_back does.  tail_info: see ops.lowrank_fit_back."""
    fits = ops.lowrank_fit_back_group(p["ctx"], d, e, Q, vecs, ss, tail_info)
    return [_assemble(m["N"], m["K"], m["rank"], m["t"], ss[b] if ss is not None else None, f["U"], f["Ut"], f["L"], f["info"],
                      m["UO"], m["Ct"], m["info_o"]) for b, (m, f) in enumerate(zip(p["members"], fits))]


def fit_error_matrix_back(p, d, e, Q, vecs, s, tail_info=None):
    """Second half of fit_error_matrix: (d, e, Q) = the tridiagonalised Rayleigh-Ritz matrix of this layer; returns what
    fit_error_matrix returns, from the same arithmetic."""
    fit = ops.lowrank_fit_back(p["ctx"], d, e, Q, vecs, s=s, tail_info=tail_info)
    return _assemble(p["N"], p["K"], p["rank"], p["t"], s, fit["U"], fit["Ut"], fit["L"], fit["info"], p["UO"], p["Ct"],
                     p["info_o"])


def outlier_columns(scaler_row, max_cols=MAX_OUTLIER_COLS, rank=None):
    """((indices of the outlier columns padded to a multiple of 8, number of real outliers) or None, spread of
    the remaining scales).  One host synchronisation; the driver precomputes the same quantities for a
    whole block at once (prune.block_stats).  rank: the adapter rank of the layer (None: not limiting)."""
    vals, idx = ops.outlier_stats(scaler_row, max_cols + 1)
    host = vals.cpu().tolist()
    k = min(max_cols + 1, scaler_row.numel())
    return _outliers_from_host(host[2:2 + k], host[1], idx[:k], rank)


def _outliers_from_host(top_vals, median, top_idx, rank=None):
    """top_vals: the largest scales in decreasing order (one more than may be listed)."""
    median = max(median, 1e-30)
    t_always = sum(1 for v in top_vals[:-1] if v > OUTLIER_RATIO_ALWAYS * median)
    t = sum(1 for v in top_vals[:-1] if v > OUTLIER_RATIO * median)
    if rank is not None and t > max(int(rank) // 16, t_always):
        t = t_always
    t_pad = min(_round_up8(t), (len(top_vals) - 1) // 8 * 8)
    t = min(t, t_pad)
    rest = top_vals[t] / median if t < len(top_vals) else 1.0
    # a split-off column leaves behind what the bf16 rounding of its exact part could not hold, about 2^-8 of it: for
    # scales thousands of times the median that remainder is itself an outlier column of the matrix the fit sees
    if t:
        rest = max(rest, top_vals[0] / median / 256.0)
    return ((top_idx[:t_pad].contiguous(), t) if t else None), max(rest, 1.0)


def lora_rank(shape, rank_ratio, lora_tile_size=None):
    """lora.py:174-181: int(ratio * min(shape)) padded up to a multiple of the tile size."""
    rank = int(rank_ratio * min(shape))
    if lora_tile_size is not None:
        residue = rank % lora_tile_size
        if residue != 0:
            rank = rank + (lora_tile_size - residue)
        assert rank % lora_tile_size == 0
    return rank


def quantize_lora(model, bitwidth=8, lora_tile_size=256):
    """Group-quantise every adapter in place and mark the layers (lora.py:56-101).  One shared
    Quantizer(block_quantization=True) instance is attached to every linear as `lora_quantizer`."""
    quantizer = AutoQuantizer("weight", num_bits=bitwidth, block_quantization=True, block_dim=lora_tile_size)
    layers = get_layers_list(model)
    progress_bar = tqdm.tqdm(range(len(layers)))
    for i in progress_bar:
        subset = find_layers(layers[i])
        for name in subset:
            progress_bar.set_description(f"Layer {i} - Quantizing LoRA for {name}")
            lin = subset[name]
            for attr in ("lora_left", "lora_right"):
                p = getattr(lin, attr)
                if not p.is_cuda:
                    raise RuntimeError("quantize_lora needs the adapters on the GPU (slim_amd has no CPU path); "
                                       "move the model to the device first")
                # same asserts and state as Quantizer.quantize_weight + dequantize_absmax
                q = quantizer.quantize_weight(p.data)
                p.data = quantizer.dequantize_absmax(q).to(lin.weight.dtype)
            lin.lora_quantizer = quantizer


class LoraContext:
    """State of one layer between the three phases of add_lora (prepare -> fit -> finish)."""

    def tensors(self):
        out = [self.W, self.W_mask, self.s, self.E, self.Et]
        if getattr(self, "fit", None) is not None:
            out += list(self.fit)
        if getattr(self, "deflated", None) is not None:
            out += list(self.deflated)
        if self.outliers is not None:
            out.append(self.outliers[0])
        return [t for t in out if isinstance(t, torch.Tensor)]

    def touch(self):
        """Tell the caching allocator that the current stream uses this layer's tensors (the phases of one
        layer may run on different streams)."""
        st = torch.cuda.current_stream()
        for t in self.tensors():
            t.record_stream(st)


# Adam iterations of the adapter re-fit after pruning (the reference's default, lora.py:11)
PRUNE_LORA_ITERS = int(os.environ.get("SLIM_PRUNE_LORA_ITERS", "1000"))


def prune_and_optimize_lora(L, R, num_iters=1000, lr_end_factor=1e-4):
    """Prune L [K, r] 2:4 along K and re-fit L and R [r, N] to the product they had (lora.py:7-53).  L and R (bf16, on
    the GPU) are updated in place; returns the bool mask of the pruned entries of L.  The mask marks the two SMALLEST
    signed values of every four, as the reference's prune_nm call on L^T does.  r must be a multiple of 8 (callers
    pad with zero columns, which stay zero)."""
    if L.dtype != torch.bfloat16 or R.dtype != torch.bfloat16 or not L.is_cuda:
        raise RuntimeError("prune_and_optimize_lora expects bf16 adapters on the GPU")
    K, r = L.shape
    N = R.shape[1]
    if r % 8 or K % 4:
        raise ValueError("prune_and_optimize_lora: the rank must be a multiple of 8 and K a multiple of 4")
    Lc, Rc = L.contiguous(), R.contiguous()
    _, T16, _ = ops.gemm_nt(Lc, ops.transpose(Rc), out_bf16=True)                 # torch.matmul(L, R) in bf16 (:25)
    mask = ops.prune_nm_f32(Lc.t().float().contiguous(), 2, 4).t().contiguous()   # :27
    Lc = Lc.masked_fill(mask, 0)
    L32, R32 = Lc.float(), Rc.float()
    ops.lora_prune_optimize(L32, R32, mask, T16.float(), num_iters, 1e6 / min(K, N) ** 2, lr_end_factor)
    L.copy_(L32.to(torch.bfloat16))
    R.copy_(R32.to(torch.bfloat16))
    return mask


def add_lora_prepare(module, W_mask, rank_ratio=0.01, slim_lora=False, activations=None, quantizer=None, prune_lora=False,
                     separate_lora=True, lora_tile_size=None, quantize_first=False, scale_important_weights=False,
                     E_out=None, Et_out=None):
    """First phase of add_lora (lora.py:105-169): argument checks, first quantisation, error matrix E (and its
    transpose) -- written into E_out / Et_out when given, so that same-shape layers of a block sit in one
    strided buffer for the batched fit."""
    if prune_lora and not separate_lora:
        raise NotImplementedError("prune_lora needs separate adapters")          # lora.py:190-191
    W = module.weight.data
    if W.dtype != torch.bfloat16 or not W.is_cuda:
        raise RuntimeError("add_lora expects a bf16 weight on the GPU")
    c = LoraContext()
    c.module, c.quantizer, c.W_mask, c.separate = module, quantizer, W_mask, bool(separate_lora)
    c.prune_lora, c.lora_left_mask = bool(prune_lora), None
    c.W = W.contiguous()
    N, K = c.W.shape
    # lora.py:132-138: the 1 % of input channels with the largest scaler_row * sum |W| are quantised at twice their size
    c.important = ops.important_columns(c.W, activations.scaler_row) if (scale_important_weights and quantizer is not None) else None
    c.bits = quantizer.num_bits if quantizer is not None else 4
    has_zero = getattr(activations, "has_zero", None)       # precomputed by the driver (one sync per block)
    if has_zero is None:
        has_zero = ops.any_zero(activations.scaler_row)
    use_slim = bool(slim_lora) and not has_zero                                 # lora.py:139
    c.s = ops.sqrt(activations.scaler_row) if use_slim else None

    # first quantisation and error matrix  (lora.py:146-156 / :158-169).  quantize_first=False quantises the
    # masked weight instead: a pruned weight quantises to 0 either way, so only the scale differs.
    Wq_src = c.W
    if quantizer is not None and not quantize_first:
        Wq_src = c.W.clone()
        Wq_src[W_mask] = 0
    # the one first quantisation that ignores the important columns: naive branch, mask first (lora.py:167)
    imp_first = c.important if (use_slim or quantize_first) else None
    if quantizer is not None and quantizer.block_quantization:
        # tiled group quantiser (quantization.py:139-148): its dequantised values are a bf16 tensor
        Wdeq = _tiled_fake_quant(Wq_src, quantizer, imp_first)
        c.E = ops.error_matrix_deq(c.W, Wdeq, W_mask, c.s, out=E_out)
        del Wdeq
    else:
        sf1 = _weight_scale(Wq_src, quantizer).reshape(1) if quantizer is not None else None
        _, c.E = ops.error_matrix(c.W, W_mask, c.s, sf1, c.bits, out=E_out, col_flags=imp_first,
                                  deq_bf16=quantizer is not None and not quantizer.slim_quant)
    c.Et = ops.transpose(c.E, out=Et_out)

    rank = lora_rank((N, K), rank_ratio, lora_tile_size)
    c.rank = min(rank, min(N, K))      # S[:rank] of a min(N, K)-long spectrum (lora.py:184)
    if use_slim:
        outliers, spread = getattr(activations, "outliers", "unset"), getattr(activations, "scale_spread", None)
        if spread is None or isinstance(outliers, str):      # not precomputed by the driver
            outliers, spread = outlier_columns(activations.scaler_row, rank=c.rank)
    else:
        outliers, spread = None, 1.0
    c.outliers, c.spread = outliers, spread
    c.level_hint = int(getattr(activations, "fit_level_hint", 0) or 0)
    c.cond_floor = int(getattr(activations, "fit_cond_floor", 0) or 0)       # > 0: this fit is being repeated
    c.fit = None
    c.deflated = None
    return c


def prepare_fusable(module, quantizer, slim_lora, prune_lora, separate_lora, quantize_first, scale_important_weights, prune_n):
    """True when add_lora_prepare_fused covers this layer's recipe: an N:M (or precomputed) mask, no quantiser or a
    per-tensor one applied to the dense weight, separate unpruned adapters."""
    W = module.weight.data
    return (W.dtype == torch.bfloat16 and W.is_cuda and W.shape[0] % 8 == 0 and W.shape[1] % 8 == 0
            and (prune_n > 0 or hasattr(module.weight, "mask"))
            and (quantizer is None or (not quantizer.block_quantization and quantize_first))
            and separate_lora and not prune_lora and not scale_important_weights)


def add_lora_prepare_fused(module, prune_n, prune_m, rank_ratio, slim_lora, activations, quantizer, lora_tile_size=None,
                           W_mask=None):
    """add_lora_prepare and add_lora_deflate with the Wanda mask before them as ONE library call (ops.lora_prepare): the
    statistics' square root, the N:M mask (unless W_mask is given), the first scale search, the error matrix and its
    transpose, and the split-off of the outlier columns.  activations must carry has_zero / outliers / scale_spread
    (prune.block_stats).  Returns the LoraContext that add_lora_fit* and add_lora_finish take."""
    c = LoraContext()
    c.module, c.quantizer, c.separate = module, quantizer, True
    c.prune_lora, c.lora_left_mask = False, None
    c.W = module.weight.data.contiguous()
    N, K = c.W.shape
    c.bits = quantizer.num_bits if quantizer is not None else 4
    use_slim = bool(slim_lora) and not activations.has_zero                     # lora.py:139
    rank = lora_rank((N, K), rank_ratio, lora_tile_size)
    c.rank = min(rank, min(N, K))
    c.outliers, c.spread = (activations.outliers, activations.scale_spread) if use_slim else (None, 1.0)
    deflate = c.outliers if fit_plan(N, K, c.rank, c.outliers)[3] else None
    mode = 0 if quantizer is None else (1 if quantizer.slim_quant else 2)
    out = ops.lora_prepare(c.W, activations.scaler_row, prune_n, prune_m, W_mask, mode, c.bits, use_slim, deflate)
    c.W_mask, c.E, c.Et = out["mask"], out["E"], out["Et"]
    c.s = out["s"] if use_slim else None
    c.level_hint = int(getattr(activations, "fit_level_hint", 0) or 0)
    c.cond_floor = int(getattr(activations, "fit_cond_floor", 0) or 0)       # > 0: this fit is being repeated
    c.fit = None
    c.deflated = out["deflated"]
    return c


def add_lora_deflate(c):
    """Between the first and the second phase, optional: split the outlier columns of the error matrix off now (the fit's
    front half does it otherwise), so that a driver can run this host-heavy step of its layers side by side."""
    N, K = c.E.shape
    if fit_plan(N, K, c.rank, c.outliers)[3]:
        c.deflated = deflated_pair(c.E, c.Et, c.outliers)


def planned_fit_l(module, rank_ratio, lora_tile_size, slim_lora, activations):
    """Size of the eigenproblem add_lora's fit will pose for this layer (needs activations.has_zero / .outliers), or None
    when the shared eigen-solver does not take it."""
    N, K = module.weight.shape
    rank = min(lora_rank((N, K), rank_ratio, lora_tile_size), min(N, K))
    use_slim = bool(slim_lora) and not activations.has_zero
    l = fit_plan(N, K, rank, activations.outliers if use_slim else None)[2]
    return l if (8 <= l <= 2048 and l < min(N, K) and not FIT_LIBRARY_EIGH) else None


def planned_fit_group(module, rank_ratio, lora_tile_size, slim_lora, activations):
    """(key, l) of a layer before it is prepared: layers with equal keys can run their fits in lockstep around the shared
    eigen-solver (same short dimension, same plan of the randomized fit, same Newton-Schulz depth -- the shapes themselves
    may differ), l is the size of the eigenproblem; None when the shared eigen-solver does not take the layer."""
    l = planned_fit_l(module, rank_ratio, lora_tile_size, slim_lora, activations)
    if l is None:
        return None
    N, K = module.weight.shape
    rank = min(lora_rank((N, K), rank_ratio, lora_tile_size), min(N, K))
    use_slim = bool(slim_lora) and not activations.has_zero
    rf, over, _, t = fit_plan(N, K, rank, activations.outliers if use_slim else None)
    spread = activations.scale_spread if use_slim else 1.0
    if FIT_CHOLESKY_QR or (t == 0 and rf != _round_up8(rank)):
        return ("alone", id(module)), l            # modes and ranks that do not run as a mixed-shape group
    shape = (N, K) if os.environ.get("SLIM_GROUP_BY_SHAPE", "0") == "1" else None
    hint = int(getattr(activations, "fit_level_hint", 0) or 0)
    floor = int(getattr(activations, "fit_cond_floor", 0) or 0)
    # (a filtered fit sizes its orthonormalisations from its filter plan, not from the hint's level.  Everything that shapes a
    # member's arithmetic is part of the key, so that a layer's result does not depend on which layers share its chain:
    # several ranks dealing the layers of a block among themselves produce what one rank does)
    plan = filter_plan([hint], floor)
    return (min(N, K), rf, over, filter_levels(plan[1]) if plan else fit_ns_steps(spread, hint, floor), plan[1] if plan else 0, shape), l


def fit_keep(c):
    """Directions the fit of this layer's context must deliver: the adapter rank minus the split-off outlier columns."""
    N, K = c.E.shape
    return c.rank - fit_plan(N, K, c.rank, c.outliers)[3]


def add_lora_fit_front_group(ctxs, M_out, sketch_split=True):
    """Second phase, first half, for a group of layers with equal planned_fit_group keys (see
    fit_error_matrix_front_group)."""
    for c in ctxs:
        c.touch()
    floor = max(c.cond_floor for c in ctxs)
    plans = [filter_plan([c.level_hint], floor) for c in ctxs]
    if len({(filter_levels(p[1]), p[1]) if p else fit_ns_steps(c.spread, c.level_hint, floor) for c, p in zip(ctxs, plans)}) != 1:
        raise ValueError("the layers of a group must share their Newton-Schulz depth")
    ctxs[0].pending = fit_error_matrix_front_group([c.E for c in ctxs], [c.Et for c in ctxs], [c.rank for c in ctxs],
                                                   [c.outliers for c in ctxs], ctxs[0].spread, M_out, sketch_split,
                                                   ctxs[0].level_hint, [c.deflated for c in ctxs], [c.level_hint for c in ctxs], floor)


def add_lora_fit_back_group(ctxs, d, e, Q, vecs, tail_info=None):
    """Second phase, second half, for the group add_lora_fit_front_group started."""
    for c in ctxs:
        c.touch()
    ss = [c.s if c.separate else None for c in ctxs]
    fits = fit_error_matrix_back_group(ctxs[0].pending, d, e, Q, vecs, ss, tail_info)
    ctxs[0].pending = None
    for c, f in zip(ctxs, fits):
        c.fit = f


def add_lora_fit_front(c, M_out, sketch_split=True):
    """Second phase, first half (see fit_error_matrix_front).  sketch_split=False: the caller runs several fits side by
    side, which fills the chip without cutting the sketch products along their reduction."""
    c.touch()
    c.pending = fit_error_matrix_front(c.E, c.Et, c.rank, c.outliers, c.spread, M_out, sketch_split, c.level_hint, c.deflated,
                                       c.cond_floor)


def add_lora_fit_back(c, d, e, Q, vecs, tail_info=None):
    """Second phase, second half (tail_info: see ops.lowrank_fit_back)."""
    c.touch()
    c.fit = fit_error_matrix_back(c.pending, d, e, Q, vecs, c.s if c.separate else None, tail_info)
    c.pending = None


def add_lora_fit(ctxs):
    """Second phase: the low-rank fit (lora.py:172-195), one context after the other."""
    for c in ctxs:
        c.touch()
        # merged adapters (separate_lora=False) divide the product, not the factor, by sqrt(scaler_row); pruned
        # adapters are pruned and re-fitted before that division (lora.py:185-195)
        scaled_in_fit = c.separate and not c.prune_lora
        c.fit = fit_error_matrix(c.E, c.Et, c.rank, c.s if scaled_in_fit else None, c.outliers, c.spread, c.level_hint,
                                 c.cond_floor)
        if c.prune_lora:
            U, Ut, L, info = c.fit
            c.lora_left_mask = prune_and_optimize_lora(L, Ut, PRUNE_LORA_ITERS)
            U = ops.transpose(Ut)
            if c.s is not None:
                ops.scale_rows(L, c.s)
            c.fit = (U, Ut, L, info)


def add_lora_finish(c, prescale=False):
    """Third phase (lora.py:199-209): residual weight W - R^T L^T, masked, second quantisation; sets
    module.lora_left [K, r], module.lora_right [r, N], module.weight.data.  Returns the second scaling
    factor (fp32 (1, 1)) or None; it is also left in quantizer.scaling_factor like the reference does.
    prescale=True (the driver, slim/prune.py:296-301): the adapters are stored multiplied by sqrt(rank) and
    module.lora_rank is set."""
    c.touch()
    U, lora_right, lora_left, fit_info = c.fit
    module, quantizer, rank = c.module, c.quantizer, c.rank
    important = getattr(c, "important", None)
    if c.separate and c.lora_left_mask is None and (quantizer is None or not quantizer.block_quantization) and important is None:
        # separate adapters, per-tensor quantiser or none: the whole tail is one library call
        mode = 0 if quantizer is None else (1 if quantizer.slim_quant else 2)
        new_weight, sf, left, right = ops.lora_finish(U, lora_left, c.W, c.W_mask, rank, mode, c.bits, prescale)
        sf_out = None
        if quantizer is not None:
            sf_out = (sf if quantizer.slim_quant else sf.to(torch.bfloat16)).reshape(1, 1)
            quantizer.scaling_factor = sf_out
        module.lora_fit_info = fit_info
        module.lora_left = torch.nn.Parameter(left)
        module.lora_right = torch.nn.Parameter(right)
        module.weight.data = new_weight
        if prescale:
            module.lora_rank = torch.tensor(rank)
        return sf_out
    low_rank_weight = None
    if c.separate:
        new_weight = ops.lowrank_residual(U, lora_left, c.W, c.W_mask)
    else:
        # lora.py:189,197,200-201: the product U S V^T in bf16, divided by bf16(sqrt(scaler_row)) column-wise
        _, low_rank_weight, _ = ops.gemm_nt(U, lora_left, out_bf16=True)
        if c.s is not None:
            low_rank_weight /= c.s.reshape(1, -1).to(torch.bfloat16)
        new_weight = c.W - low_rank_weight
        new_weight[c.W_mask] = 0
    new_weight, sf_out = requantize_weight(new_weight, quantizer, c.bits, important)
    module.lora_fit_info = fit_info
    if not c.separate:
        # lora.py:213: the adapters are folded back into the (then dense) weight
        module.weight.data = (new_weight.float() + low_rank_weight.float()).to(torch.bfloat16).contiguous()
        return sf_out
    if lora_left.shape[1] != rank:       # the factors are padded to a multiple of 8 columns
        lora_left, lora_right = lora_left[:, :rank], lora_right[:rank]
    module.lora_left = torch.nn.Parameter(lora_left).to(torch.bfloat16).contiguous()
    module.lora_right = torch.nn.Parameter(lora_right).to(torch.bfloat16).contiguous()
    module.weight.data = new_weight.to(torch.bfloat16).contiguous()
    if c.lora_left_mask is not None:
        module.lora_left_mask = c.lora_left_mask[:, :rank].contiguous()              # lora.py:210-211
    if prescale:
        module.lora_rank = torch.tensor(module.lora_left.shape[1])
        module.lora_left = torch.nn.Parameter(module.lora_left * torch.sqrt(module.lora_rank))
        module.lora_right = torch.nn.Parameter(module.lora_right * torch.sqrt(module.lora_rank))
    return sf_out


def add_lora(module, W_mask, rank_ratio=0.01, slim_lora=False, activations=None, quantizer=None, prune_lora=False,
             separate_lora=True, lora_tile_size=None, quantize_first=False, scale_important_weights=False):
    """Fit low-rank adapters to the compression error of `module` and re-quantise the residual weight
    (lora.py:105-213).  Sets module.lora_left [K, r], module.lora_right [r, N] and module.weight.data.

    Implemented: quantizer None, per-tensor (SLiM-Quant / abs-max) or tiled group abs-max, slim_lora on or off,
    quantize_first either way, adapters kept separate, merged back or pruned, scale_important_weights.
    The driver runs the three phases separately so that same-shape layers share one batched fit.
    """
    c = add_lora_prepare(module, W_mask, rank_ratio, slim_lora, activations, quantizer, prune_lora, separate_lora,
                         lora_tile_size, quantize_first, scale_important_weights)
    add_lora_fit([c])
    add_lora_finish(c)


def requantize_weight(new_weight, quantizer, bits, important=None):
    """The second quantisation of add_lora (lora.py:202-204): (deq(quant(new_weight)) in bf16, the scaling factor (1, 1) or
    None), which is also left in quantizer.scaling_factor like the reference does.  important: bool [K] flags of the
    columns quantised at twice their size (scale_important_weights), or None."""
    if quantizer is None:
        return new_weight, None
    if quantizer.block_quantization:
        return _tiled_fake_quant(new_weight, quantizer, important), None
    sf2 = _weight_scale(new_weight, quantizer)             # the threshold comes from the unscaled weight (:184-192)
    if important is not None:
        # codes of the doubled columns, dequantised values halved (quantization.py:192, :215): a power of two commutes
        # with the rounding to bf16
        new_weight = ops.fake_quant_pt(ops.scale_columns(new_weight, important, 2.0), sf2.reshape(1), bits)
        ops.scale_columns(new_weight, important, 0.5, out=new_weight)
    else:
        new_weight = ops.fake_quant_pt(new_weight, sf2.reshape(1), bits)
    sf_out = (sf2 if quantizer.slim_quant else sf2.to(torch.bfloat16)).reshape(1, 1)
    quantizer.scaling_factor = sf_out
    return new_weight, sf_out


def _tiled_fake_quant(W, quantizer, important=None):
    """deq(quant(W)) under the tiled group quantiser (Quantizer.quantize_weight block branch + dequantize_absmax,
    quantization.py:139-148, :203-217) without touching the quantizer's state: layers are compressed concurrently.
    important (bool [K] or None): those columns are doubled after the group scales have been taken and halved after
    dequantisation (:146-148, :215)."""
    assert W.shape[0] % quantizer.block_dim == 0 and W.shape[1] % quantizer.block_dim == 0, \
        "Input matrix size is not divisible by block size"
    if quantizer.slim_quant:
        raise NotImplementedError("SLiM-Quant is not supported for block quantization")
    if important is None:
        return ops.group_fake_quant(W.contiguous(), quantizer.block_dim, quantizer.num_bits)
    W = W.contiguous()
    alphas = ops.group_absmax(W, quantizer.block_dim)
    codes = ops.group_quantize(ops.scale_columns(W, important, 2.0), alphas, quantizer.num_bits)
    deq = ops.group_dequantize(codes, alphas, quantizer.num_bits)
    return ops.scale_columns(deq, important, 0.5, out=deq)


def _weight_scale(W, quantizer):
    """Per-tensor scaling factor of `W` under the quantizer's policy, as a 0-d fp32 device tensor."""
    hist = ops.hist_abs(W)
    if quantizer.slim_quant:
        bins = max(512, min(W.numel() // 1000, 20000))
        _, _, sf, _ = ops.quant_search(hist, bins, quantizer.num_bits)
    else:
        _, sf = ops.absmax_scale(hist, quantizer.num_bits)
    return sf
