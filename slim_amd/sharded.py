"""A decoder block's linears compressed by several GPUs together: every linear is cut along its LONG dimension (row blocks
of W when out_features >= in_features, column blocks otherwise) and every rank works on its block of every linear.
The reference has no distributed code; this is new design (torch.distributed: RCCL on the GPUs, gloo staged through the
host in rehearsals and tests).

Per linear, with Dm its short dimension and E the error matrix (slim/lora.py:146-156):
  local       saliency + 2:4 mask, error matrix and the Dm x Dm Gram matrix of the block's part of E      (no exchange)
  all-reduce  the |W| histogram (128 KiB of counts): every rank then runs the identical scale search      (twice)
  reduce      the Gram contributions (fp32 Dm x Dm) to the linear's OWNER, which runs the subspace iteration
              (the linears of a block are dealt round-robin to the owners)
  broadcast   the basis P (bf16 Dm x l) from the owner
  all-reduce  the block's share Z^T Z of the l x l Rayleigh-Ritz matrix; the owner solves the eigenproblem (all
              problems a rank owns in one batched chain) and broadcasts eigenvectors and eigenvalues
  local       the block's rows of the factors, the residual weight, re-quantisation
  all-gather  the compressed weight blocks and the blocks of the sharded factor: every rank ends with the whole
              compressed linear, as the next forward pass needs it
Masks and first-pass codes are those of a single-GPU run bit for bit (same saliency, same integer histogram).  The Gram
matrix is a sum of per-rank fp32 partial sums instead of one pass, so the factors agree with a single-GPU run to the
accuracy of the fit (reconstruction error within 1e-3 of the truncated SVD's), not to the bit.
"""
import torch

from . import ops, parallel
from . import lora as _lora

ALIGN = 128      # shard boundaries: multiples of 128 (2:4 groups, 8-aligned MFMA reductions, 128-row packing tiles)


def shard_bounds(n, world, align=ALIGN):
    """world contiguous [lo, hi) pieces of range(n) with boundaries at multiples of `align`, sizes as equal as the
    alignment allows (the first n / align % world pieces hold one aligned block more)."""
    blocks = n // align
    if blocks * align != n:
        raise ValueError(f"dimension {n} is not a multiple of {align}")
    base, extra = divmod(blocks, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + (base + (1 if r < extra else 0)) * align
        out.append((lo, hi))
        lo = hi
    return out


def supported(items, world, prune_n, prune_m, quantizer, lora_rank, slim_lora=True, prune_lora=False, separate_lora=True,
              quantize_first=True, tile=None):
    """Whether compress_linears_sharded takes this block: the SLiM recipe (N:M mask from the saliency, per-tensor
    quantiser, quantise first, separate unpruned adapters) on linears whose long dimension gives every rank at least one
    aligned block.  Otherwise the driver keeps whole linears on their owners."""
    if not (lora_rank > 0. and separate_lora and not prune_lora and quantize_first):
        return False
    if prune_n == 0 or quantizer is None or quantizer.block_quantization:
        return False
    for _, lin, wrapped in items:
        N, K = lin.weight.shape
        if hasattr(lin.weight, "mask") or getattr(wrapped, "has_zero", None) is None:
            return False
        long_dim, Dm = (N, K) if K <= N else (K, N)
        if long_dim % ALIGN or long_dim // ALIGN < world or Dm % 8 or K % prune_m:
            return False
        rank = min(_lora.lora_rank((N, K), lora_rank, tile), min(N, K))
        use_slim = bool(slim_lora) and not wrapped.has_zero
        rf, over, l, t = _lora.fit_plan(N, K, rank, wrapped.outliers if use_slim else None)
        if not (8 <= l < Dm and l <= 2048 and rf <= l and rf == _lora._round_up8(rank - t)):
            return False
    return True


def _t(x):
    return ops.transpose(x) if x.dtype == torch.bfloat16 else x.t().contiguous()


def _gather_blocks(x, bounds, dim, group=None):
    """x: this rank's block along `dim` of a 2-D tensor (block sizes given by bounds); returns the whole tensor."""
    sizes = [hi - lo for lo, hi in bounds]
    mx = max(sizes)
    xt = x.contiguous() if dim == 0 else _t(x)
    if xt.shape[0] != mx:
        pad = torch.zeros(mx, xt.shape[1], dtype=x.dtype, device=x.device)
        pad[:xt.shape[0]] = xt
        xt = pad
    parts = parallel.all_gather(xt, group)                  # [world, mx, cols]
    if min(sizes) == mx:
        full = parts.reshape(-1, xt.shape[1])
    else:
        full = torch.cat([parts[r, :sizes[r]] for r in range(len(bounds))], 0)
    return full.contiguous() if dim == 0 else _t(full.contiguous())


class _Job:
    """State of one linear between the phases."""


def _scale(hist, numel, quantizer):
    """Per-tensor scaling factor (0-d fp32 device tensor) from the summed |W| histogram under the quantizer's policy
    (slim/quantization/quantization.py:34-62 for SLiM-Quant, :160-166 for abs-max)."""
    if quantizer.slim_quant:
        bins = max(512, min(numel // 1000, 20000))
        return ops.quant_search(hist, bins, quantizer.num_bits)[2]
    return ops.absmax_scale(hist, quantizer.num_bits)[1]


def _deflate(j, outliers, dev, group):
    """Outlier columns of E (saliency scale far above the median) are represented exactly before the fit, as in
    lora.fit_error_matrix: with Q_O an orthonormal basis of their span, E = Q_O (Q_O^T E) + E'.  The columns are
    collected from the blocks, the small basis is formed on every rank from identical input (identical result), and every
    rank deflates its block.  Leaves j.E_s / j.Et_s = the block of E', j.defl_rows = the block's rows of the exact part's
    factor on the sharded side, j.defl_dm = the whole factor on the other side."""
    cols, t = outliers
    tp = cols.numel()
    N, K = j.N, j.K
    EO = torch.zeros(N, tp, dtype=torch.float32, device=dev)
    if j.right:
        EO[j.lo:j.hi] = torch.index_select(j.E_s, 1, cols).float()
    else:
        own = ((cols >= j.lo) & (cols < j.hi)).float()
        local = (cols - j.lo).clamp(0, j.hi - j.lo - 1)
        EO.copy_(torch.index_select(j.E_s, 1, local).float() * own.reshape(1, -1))
    parallel.all_reduce(EO, group)              # every entry comes from exactly one rank
    # the gathered columns are bf16 values (every entry comes from one rank's block of E): the single-GPU basis routine
    # on them gives every rank the basis a single GPU computes
    UO, UOt = ops.outlier_basis(EO.t().contiguous().to(torch.bfloat16), torch.arange(tp, dtype=torch.int32, device=dev), t)
    j.info_o = torch.zeros(64, dtype=torch.int32, device=dev)
    if j.right:
        # C = Q_O^T E [tp, K]: a sum over the row blocks
        C32 = ops.gemm_nt(UOt[:, j.lo:j.hi].contiguous(), j.Et_s, out_f32=True)[0]
        parallel.all_reduce(C32, group)
        Ct = C32.to(torch.bfloat16).t().contiguous()                   # [K, tp]
        UO_s = UO[j.lo:j.hi].contiguous()
        j.E_s = ops.lowrank_residual(UO_s, Ct, j.E_s, None)
        j.defl_rows = UO_s
        j.defl_dm = ops.scale_rows(Ct.clone(), j.s) if j.s is not None else Ct
    else:
        _, C, Ct_s = ops.gemm_nt(UOt, j.Et_s, out_bf16=True, out_t=True)      # [tp, K_s], [K_s, tp]
        j.E_s = ops.lowrank_residual(UO, Ct_s, j.E_s, None)
        j.defl_rows = ops.scale_rows(Ct_s.clone(), j.s_s) if j.s_s is not None else Ct_s
        j.defl_dm = UO
    j.Et_s = ops.transpose(j.E_s)


def compress_linears_sharded(items, dev, prune_n, prune_m, quantizer, lora_rank, slim_lora=True, tile=None, callback=None,
                             describe=None, group=None):
    """The adapters branch of prune.compress_linears_unchecked for a block whose linears are all cut into blocks over the
    ranks (see the module docstring).  items: list of (name, nn.Linear, WrappedGPT) with prune.block_stats applied,
    identical on every rank.  Leaves on every rank what the single-GPU path leaves: weight.data, lora_left, lora_right,
    lora_rank, scaling_factor, the adapter hook, and lora_fit_info (int32 [64], maximised over the ranks: word 0 = a
    status word of the fit was non-zero somewhere, last word = conditioning level the fit asks to be repeated with)."""
    from .prune import add_lora_hook
    rank, world = parallel.world_info(group)
    jobs = []
    # ---- phase 1: local elementwise stages, Gram contributions, iteration on the owners ---------------------------------
    for idx, (name, lin, wrapped) in enumerate(items):
        if describe:
            describe(name)
        j = _Job()
        jobs.append(j)
        W = lin.weight.data
        N, K = W.shape
        j.N, j.K, j.right = N, K, K <= N
        j.owner = idx % world
        long_dim, j.Dm = (N, K) if j.right else (K, N)
        j.bounds = shard_bounds(long_dim, world)
        j.lo, j.hi = j.bounds[rank]
        j.rank_ = min(_lora.lora_rank((N, K), lora_rank, tile), min(N, K))
        use_slim = bool(slim_lora) and not wrapped.has_zero                        # lora.py:139
        s_mask = ops.sqrt(wrapped.scaler_row)                                       # prune.py:264-265
        j.s = s_mask if use_slim else None
        outliers = wrapped.outliers if use_slim else None
        j.rf, j.over, j.l, j.t = _lora.fit_plan(N, K, j.rank_, outliers)
        j.cond = _lora.fit_ns_steps(wrapped.scale_spread if use_slim else 1.0)
        j.bits = quantizer.num_bits
        if j.right:
            j.W_s = W[j.lo:j.hi].contiguous()
            j.s_s, sm_s = j.s, s_mask
        else:
            j.W_s = W[:, j.lo:j.hi].contiguous()
            j.s_s = j.s[j.lo:j.hi].contiguous() if j.s is not None else None
            sm_s = s_mask[j.lo:j.hi].contiguous()
        j.mask_s = ops.mask_nm(j.W_s, sm_s, prune_n, prune_m)
        hist = torch.zeros(32768, dtype=torch.int32, device=dev)
        ops.hist_abs(j.W_s, hist)
        parallel.all_reduce(hist, group)
        sf1 = _scale(hist, N * K, quantizer)
        _, j.E_s = ops.error_matrix(j.W_s, j.mask_s, j.s_s, sf1.reshape(1), j.bits, deq_bf16=not quantizer.slim_quant)
        j.Et_s = ops.transpose(j.E_s)
        j.defl_rows = j.defl_dm = j.info_o = None
        if j.t:
            _deflate(j, outliers, dev, group)
        # Gram contribution over the sharded dimension: E_s^T E_s [K, K] (row blocks) or E_s E_s^T [N, N] (column blocks)
        op = j.Et_s if j.right else j.E_s
        Cp = ops.gemm_nt(op, op, out_f32=True)[0]
        parallel.reduce(Cp, j.owner, group)
        if rank == j.owner:
            j.P, _, j.info = ops.gramfit_iterate(Cp.to(torch.bfloat16), j.l, _lora.fit_iters(j.l), seed=_lora.FIT_SEED, cond_e=j.cond)
        else:
            j.P = torch.empty(j.Dm, j.l, dtype=torch.bfloat16, device=dev)
            j.info = torch.zeros(64, dtype=torch.int32, device=dev)
        del Cp
    # ---- phase 2: the bases to everyone, projections, Rayleigh-Ritz matrices ----------------------------------------------
    for j in jobs:
        parallel.broadcast(j.P, j.owner, group)
        # right: Z_s = E_s P [rows, l];  left: Z_s = E_s^T P [K_s, l]
        j.Z, _, j.M = ops.gramfit_project(j.E_s if j.right else j.Et_s, ops.transpose(j.P))
        parallel.all_reduce(j.M, group)
    # ---- phase 3: the eigenproblems, every owner its own in one batched chain per size -------------------------------------
    mine = {}
    for i, j in enumerate(jobs):
        if j.owner == rank:
            mine.setdefault(j.l, []).append(i)
    for l, idxs in sorted(mine.items()):
        Mb = torch.stack([jobs[i].M for i in idxs]).contiguous()
        d, e, Q = ops.tridiagonalize_batched(Mb)
        tail = ops.eigh_tail(d, e, Q, Mb, max(jobs[i].rf for i in idxs))
        for b, i in enumerate(idxs):
            jobs[i].M = Mb[b].contiguous()
            jobs[i].evals = d[b].contiguous()
            jobs[i].info[61] = tail[b]
    for j in jobs:
        if j.owner != rank:
            j.evals = torch.empty(j.l, dtype=torch.float32, device=dev)
        parallel.broadcast(j.M, j.owner, group)         # its last rf rows hold the wanted eigenvectors
        parallel.broadcast(j.evals, j.owner, group)
    # ---- phase 4: factors, residual, second scale, re-quantisation, gather --------------------------------------------------
    status = torch.zeros(len(jobs), 2, dtype=torch.int32, device=dev)
    for idx, (j, (name, lin, wrapped)) in enumerate(zip(jobs, items)):
        r8 = _lora._round_up8(j.rank_)
        nfit = j.rank_ - j.t
        # right: dm = lora_left [K, rf] (whole), rows = this block's rows of U;  left: dm = U [N, rf] (whole), rows = this
        # block's rows of lora_left
        dm, _, rows, _, _ = ops.gramfit_factors(j.right, j.P, j.Z, j.M, j.evals, j.rf, s=j.s if j.right else j.s_s,
                                                info=j.info, cond_e=j.cond)
        rows_full = torch.zeros(rows.shape[0], r8, dtype=torch.bfloat16, device=dev)
        dm_full = torch.zeros(j.Dm, r8, dtype=torch.bfloat16, device=dev)
        if j.t:
            rows_full[:, :j.t] = j.defl_rows[:, :j.t]
            dm_full[:, :j.t] = j.defl_dm[:, :j.t]
        rows_full[:, j.t:j.rank_] = rows[:, :nfit]
        dm_full[:, j.t:j.rank_] = dm[:, :nfit]
        U_s, L_s = (rows_full, dm_full) if j.right else (dm_full, rows_full)
        Wn_s = ops.lowrank_residual(U_s, L_s, j.W_s, j.mask_s)                      # lora.py:199-201
        hist = torch.zeros(32768, dtype=torch.int32, device=dev)
        ops.hist_abs(Wn_s, hist)
        parallel.all_reduce(hist, group)
        sf2 = _scale(hist, j.N * j.K, quantizer)
        Wq_s = ops.fake_quant_pt(Wn_s, sf2.reshape(1), j.bits)                      # lora.py:202-204
        lin.weight.data = _gather_blocks(Wq_s, j.bounds, 0 if j.right else 1, group)
        rows_all = _gather_blocks(rows_full, j.bounds, 0, group)
        U, L = (rows_all, dm_full) if j.right else (dm_full, rows_all)
        sf_out = (sf2 if quantizer.slim_quant else sf2.to(torch.bfloat16)).reshape(1, 1)
        quantizer.scaling_factor = sf_out
        lin.scaling_factor = sf_out
        lin.lora_rank = torch.tensor(j.rank_)
        sq = torch.sqrt(lin.lora_rank)
        lin.lora_left = torch.nn.Parameter(L[:, :j.rank_].contiguous() * sq)
        lin.lora_right = torch.nn.Parameter(ops.transpose(U)[:j.rank_].contiguous() * sq)
        lin.register_forward_hook(add_lora_hook)
        bad = j.info[:-3].ne(0).any()
        if j.info_o is not None:
            bad = bad | j.info_o.ne(0).any()
        status[idx, 0] = bad.to(torch.int32)
        status[idx, 1] = j.info[-1]
    parallel.all_reduce(status, group, op="max")
    for idx, (name, lin, wrapped) in enumerate(items):
        info = torch.zeros(64, dtype=torch.int32, device=dev)
        info[0] = status[idx, 0]
        info[-1] = status[idx, 1]
        lin.lora_fit_info = info
        if callback is not None:
            callback(name, lin, wrapped)
    return status
