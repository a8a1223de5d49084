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
import os

import torch

from . import ops, parallel
from . import lora as _lora

# Side streams (each fed by a host thread) for the per-linear local phases.  Off by default: between the library calls these
# phases are made of small torch operations (slices, copies, casts) whose host cost -- 30 to 45 us each here, under the
# interpreter lock -- is what bounds the step, and threads only add contention there (measured on a LLaMA-2-7B-shaped block
# on one rank: 20.5 ms in order, 27.6 ms over seven streams).
SHARD_STREAMS = int(os.environ.get("SLIM_SHARD_STREAMS", "1"))

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
    if len(bounds) == 1:
        return x.contiguous()
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


_aranges = {}


def _arange(tp, dev):
    key = (tp, str(dev))
    if key not in _aranges:
        _aranges[key] = torch.arange(tp, dtype=torch.int32, device=dev)
    return _aranges[key]


def _deflate(j, outliers, dev, group):
    """Outlier columns of E (saliency scale far above the median) are represented exactly before the fit, as in
    lora.fit_error_matrix: with Q_O an orthonormal basis of their span, E = Q_O (Q_O^T E) + E'.  The columns are
    collected from the blocks (as rows of E^T: bf16, every entry from exactly one rank), the small basis is formed on every
    rank from identical input (identical result, and the basis a single GPU computes), and every rank deflates its block
    (ops.deflate_apply: E' and its transpose in one pass).  Leaves j.E_s / j.Et_s = the block of E', j.defl_rows = the
    block's rows of the exact part's factor on the sharded side, j.defl_dm = the whole factor on the other side; the factor
    that becomes lora_left is left undivided by the saliency scales (ops.lora_assemble divides)."""
    cols, t = outliers
    tp = cols.numel()
    world = len(j.bounds)
    if j.right:
        # rows `cols` of E_s^T [K, rows_s]: this block's part of the outlier columns; the ranks' parts side by side
        EOt = _gather_blocks(torch.index_select(j.Et_s, 0, cols), j.bounds, 1, group)             # [tp, N]
    else:
        # column blocks: a column lives on one rank, the others contribute zeros
        own = ((cols >= j.lo) & (cols < j.hi))
        local = (cols - j.lo).clamp(0, j.hi - j.lo - 1)
        EOt = torch.index_select(j.Et_s, 0, local) * own.reshape(-1, 1).to(torch.bfloat16)        # [tp, N]
        if world > 1:
            EO32 = EOt.float()
            parallel.all_reduce(EO32, group)
            EOt = EO32.to(torch.bfloat16)
    UO, UOt = ops.outlier_basis(EOt, _arange(tp, dev), t)
    j.info_o = None            # (the basis routine drops a vanishing pivot's column instead of failing: nothing to report)
    if j.right:
        # C = Q_O^T E [tp, K]: a sum over the row blocks
        if world == 1:
            Ct = ops.outlier_project(j.Et_s, UOt)                      # [K, tp]
            UO_s = UO
        else:
            C32 = ops.gemm_nt(UOt[:, j.lo:j.hi].contiguous(), j.Et_s, out_f32=True)[0]
            parallel.all_reduce(C32, group)
            Ct = C32.to(torch.bfloat16).t().contiguous()
            UO_s = UO[j.lo:j.hi].contiguous()
        j.E_s, j.Et_s = ops.deflate_apply(j.E_s, UO_s, Ct)
        j.defl_rows, j.defl_dm = UO_s, Ct
    else:
        Ct_s = ops.outlier_project(j.Et_s, UOt)                        # [K_s, tp]
        j.E_s, j.Et_s = ops.deflate_apply(j.E_s, UO, Ct_s)
        j.defl_rows, j.defl_dm = Ct_s, UO


def compress_linears_sharded(items, dev, prune_n, prune_m, quantizer, lora_rank, slim_lora=True, tile=None, callback=None,
                             describe=None, group=None):
    """The adapters branch of prune.compress_linears_unchecked for a block whose linears are all cut into blocks over the
    ranks (see the module docstring).  items: list of (name, nn.Linear, WrappedGPT) with prune.block_stats applied,
    identical on every rank (including the hints wrapped.fit_level_hint and the floors wrapped.fit_cond_floor that
    prune.compress_linears sets).  Leaves on every rank what the single-GPU path leaves: weight.data, lora_left,
    lora_right, lora_rank, scaling_factor, the adapter hook, and lora_fit_info (int32 [64], maximised over the ranks:
    word 0 = a status word of the fit was non-zero somewhere, the last three = the fit's reports as slim_lowrank_fit leaves
    them: conditioning level to repeat with, measured spread, where the cut sits).

    The exchanges of a block are batched: ONE all-reduce carries the first-pass histograms of all linears, one the
    second-pass histograms, one per eigenproblem size the Rayleigh-Ritz matrices, one the status words.  The owners run the
    subspace iterations of their linears in lockstep (one batched launch sequence per class of equal size and depth,
    Chebyshev-filtered when the hints carry the previous block's measurements, as on one GPU), and every rank solves the
    small Rayleigh-Ritz stage of all linears itself, from identical input: no broadcast of eigenvectors."""
    from .prune import add_lora_hook, run_on_streams
    rank, world = parallel.world_info(group)
    n = len(items)
    jobs = [_Job() for _ in items]
    import time
    trace = os.environ.get("SLIM_SHARD_TRACE") == "1"
    t_host0 = time.perf_counter()
    marks = []

    def mark(name):
        # (debugging aid: an event on the calling stream and the host clock per phase, printed at the end -- no waiting in between)
        if trace:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append((name, ev, (time.perf_counter() - t_host0) * 1e3))
    mark("start")

    def fan_out(fn, count=n):
        """fn(i) for i < count on side streams (local work only: collectives stay on the calling thread, in one order on
        every rank); the tensors the jobs hold are handed back to the calling stream."""
        if torch.device(dev).type != "cuda" or SHARD_STREAMS <= 1:
            for i in range(count):
                fn(i)
            return

        def one(i):
            fn(i)
            return [t for j in jobs for t in vars(j).values() if isinstance(t, torch.Tensor)] if count != n else \
                   [t for t in vars(jobs[i]).values() if isinstance(t, torch.Tensor)]
        run_on_streams([lambda i=i: one(i) for i in range(count)], dev, SHARD_STREAMS)

    # ---- phase 1a: masks and |W| histograms of the blocks; one all-reduce for the block's histograms ---------------------
    hists = torch.zeros(n, 32768, dtype=torch.int32, device=dev)

    def local_mask(idx):
        name, lin, wrapped = items[idx]
        j = jobs[idx]
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
        j.outliers = wrapped.outliers if use_slim else None
        j.rf, j.over, j.l, j.t = _lora.fit_plan(N, K, j.rank_, j.outliers)
        # how the iteration is sized: as the single-GPU driver sizes it (hint of the previous block, floor of a repeat)
        j.hint = int(getattr(wrapped, "fit_level_hint", 0) or 0)
        j.floor = int(getattr(wrapped, "fit_cond_floor", 0) or 0)
        j.plan = _lora.filter_plan([j.hint], j.floor)
        j.cond = (_lora.filter_levels(j.plan[1]) if j.plan
                  else _lora.fit_ns_steps(wrapped.scale_spread if use_slim else 1.0, j.hint, j.floor))
        j.n_iter = j.plan[2] if j.plan else _lora.fit_iters(j.l)
        j.bits = quantizer.num_bits
        if j.right:
            j.W_s = W[j.lo:j.hi].contiguous()
            j.s_s, sm_s = j.s, s_mask
        else:
            j.W_s = W[:, j.lo:j.hi].contiguous()
            j.s_s = j.s[j.lo:j.hi].contiguous() if j.s is not None else None
            sm_s = s_mask[j.lo:j.hi].contiguous()
        j.mask_s = ops.mask_nm(j.W_s, sm_s, prune_n, prune_m)
        ops.hist_abs(j.W_s, hists[idx])
    if describe:
        for name, _, _ in items:
            describe(name)
    fan_out(local_mask)
    parallel.all_reduce(hists, group)
    mark("mask+hist")

    # ---- phase 1b: first scale, error matrix, outlier columns, Gram contributions to the owners ----------------------------
    def local_error(idx):
        j = jobs[idx]
        sf1 = _scale(hists[idx], j.N * j.K, quantizer)
        _, j.E_s = ops.error_matrix(j.W_s, j.mask_s, j.s_s, sf1.reshape(1), j.bits, deq_bf16=not quantizer.slim_quant)
        j.Et_s = ops.transpose(j.E_s)
        j.defl_rows = j.defl_dm = j.info_o = None
    fan_out(local_error)
    for j in jobs:
        if j.t:
            _deflate(j, j.outliers, dev, group)
    mark("error+deflate")
    # the linears an owner iterates on in lockstep: same size, same depth, same filter class
    classes = {}
    for idx, j in enumerate(jobs):
        key = (j.owner, j.Dm, j.l, j.cond, j.plan[1] if j.plan else 0, j.n_iter)
        classes.setdefault(key, []).append(idx)
    Cb = {key: torch.empty(len(members), key[1], key[1], dtype=torch.bfloat16, device=dev)
          for key, members in classes.items() if key[0] == rank}
    slot = {idx: (key, b) for key, members in classes.items() for b, idx in enumerate(members)}

    def local_gram(idx):
        # Gram contribution over the sharded dimension: E_s^T E_s [K, K] (row blocks) or E_s E_s^T [N, N] (column blocks)
        j = jobs[idx]
        op = j.Et_s if j.right else j.E_s
        key, b = slot[idx]
        if world == 1:
            ops.gemm_nt(op, op, C16=Cb[key][b])
        else:
            j.Cp = ops.gemm_nt(op, op, out_f32=True)[0]
    fan_out(local_gram)
    if world > 1:
        for idx, j in enumerate(jobs):
            parallel.reduce(j.Cp, j.owner, group)
            if rank == j.owner:
                key, b = slot[idx]
                Cb[key][b].copy_(j.Cp)
            del j.Cp

    mark("gram")
    # ---- phase 2: the owners' lockstep iterations; bases to everyone; projections; Rayleigh-Ritz matrices -----------------
    keys = list(classes)
    reports = {}

    def iterate(c):
        key = keys[c]
        owner, Dm, l, cond, fbits, n_iter = key
        members = classes[key]
        if owner == rank:
            plans = [jobs[i].plan for i in members]
            ops.set_fit_filter([p[0][0] for p in plans] if plans[0] else [], fbits)
            P, Pt, info, scal = ops.gramfit_iterate_batched(Cb.pop(key), l, n_iter, seed=_lora.FIT_SEED, cond_e=cond)
            for b, i in enumerate(members):
                jobs[i].P, jobs[i].Pt = P[b], Pt[b]
            reports[key] = (info, scal)
        else:
            for i in members:
                jobs[i].P, jobs[i].Pt = torch.empty(Dm, l, dtype=torch.bfloat16, device=dev), None
    fan_out(iterate, len(keys))
    mark("iterate")
    sizes = {}
    for idx, j in enumerate(jobs):
        sizes.setdefault((j.l, j.rf), []).append(idx)
    Ms = {k: torch.empty(len(m), k[0], k[0], dtype=torch.float32, device=dev) for k, m in sizes.items()}
    where = {i: (k, b) for k, members in sizes.items() for b, i in enumerate(members)}
    for j in jobs:
        parallel.broadcast(j.P, j.owner, group)

    def project(idx):
        j = jobs[idx]
        if j.Pt is None:
            j.Pt = ops.transpose(j.P)
        # right: Z_s = E_s P [rows, l];  left: Z_s = E_s^T P [K_s, l]
        k, b = where[idx]
        j.Z = ops.gramfit_project(j.E_s if j.right else j.Et_s, j.Pt, M32=Ms[k][b])[0]
    fan_out(project)
    for k in sizes:
        parallel.all_reduce(Ms[k], group)

    mark("project")
    # ---- phase 3: the Rayleigh-Ritz stage of every linear, on every rank from identical input ------------------------------
    size_keys = list(sizes)

    def ritz(c):
        k = size_keys[c]
        members = sizes[k]
        evals, vecs, tail = _lora.rayleigh_ritz(Ms[k], k[1], [jobs[i].rank_ - jobs[i].t for i in members],
                                                max(jobs[i].floor for i in members))
        for b, i in enumerate(members):
            jobs[i].evals, jobs[i].vecs, jobs[i].tail = evals[b], vecs[b], tail[b]
    fan_out(ritz, len(size_keys))
    mark("ritz")
    status = torch.zeros(n, 4, dtype=torch.int32, device=dev)
    for key, (info, scal) in reports.items():
        members = classes[key]
        # the owner's reports: conditioning against what the spectrum shows, measured spread, where the cut sits
        ops.gramfit_check(torch.stack([jobs[i].evals for i in members]), jobs[members[0]].rf, key[3], scal, info)
        bad = info[:, :-3].ne(0).any(1) | torch.stack([jobs[i].tail for i in members]).ne(0)
        report = torch.cat([bad.to(torch.int32).reshape(-1, 1), info[:, -3:].flip(1)], 1)
        for b, i in enumerate(members):           # (row by row: an index tensor built from a host list would wait for the GPU)
            status[i] = report[b]

    # ---- phase 4: factors, residual, second histograms (one all-reduce), re-quantisation, gather ---------------------------
    hists.zero_()

    def factors(idx):
        j = jobs[idx]
        r8 = _lora._round_up8(j.rank_)
        nfit = j.rank_ - j.t
        # right: dm = lora_left [K, rf] (whole), rows = this block's rows of U;  left: dm = U [N, rf] (whole), rows = this
        # block's rows of lora_left
        dm, _, rows, _, _ = ops.gramfit_factors(j.right, j.P, j.Z, j.vecs, j.evals, j.rf, s=j.s if j.right else j.s_s)
        if j.t == 0 and r8 == j.rf:
            rows_full, dm_full = rows, dm
            if j.rank_ != r8:
                rows_full[:, j.rank_:] = 0
                dm_full[:, j.rank_:] = 0
        elif j.t == 0:
            rows_full = torch.zeros(rows.shape[0], r8, dtype=torch.bfloat16, device=dev)
            dm_full = torch.zeros(j.Dm, r8, dtype=torch.bfloat16, device=dev)
            rows_full[:, :j.rank_] = rows[:, :nfit]
            dm_full[:, :j.rank_] = dm[:, :nfit]
        elif j.right:
            # U rows = [Q_O rows | fit], lora_left = [C^T / bf16(s) | fit] (one call: the exact part in front, zeros behind)
            rows_full, _, dm_full = ops.lora_assemble(j.defl_rows, j.defl_dm, j.s, rows, dm, j.t, j.rank_, r8)
        else:
            # U = [Q_O | fit], lora_left rows = [C^T rows / bf16(s) | fit]
            dm_full, _, rows_full = ops.lora_assemble(j.defl_dm, j.defl_rows, j.s_s, dm, rows, j.t, j.rank_, r8)
        j.rows_full, j.dm_full = rows_full, dm_full
        U_s, L_s = (rows_full, dm_full) if j.right else (dm_full, rows_full)
        j.Wn_s = ops.lowrank_residual(U_s, L_s, j.W_s, j.mask_s)                    # lora.py:199-201
        ops.hist_abs(j.Wn_s, hists[idx])
    fan_out(factors)
    parallel.all_reduce(hists, group)
    parallel.all_reduce(status, group, op="max")
    mark("factors")

    def requantise(idx):
        j = jobs[idx]
        j.sf2 = _scale(hists[idx], j.N * j.K, quantizer)
        j.Wq_s = ops.fake_quant_pt(j.Wn_s, j.sf2.reshape(1), j.bits)                # lora.py:202-204
    fan_out(requantise)
    mark("requant")
    for j in jobs:                                                                   # (collectives: one order on every rank)
        j.W_all = _gather_blocks(j.Wq_s, j.bounds, 0 if j.right else 1, group)
        j.rows_all = _gather_blocks(j.rows_full, j.bounds, 0, group)

    infos = torch.zeros(n, 64, dtype=torch.int32, device=dev)
    infos[:, 0] = status[:, 0]
    infos[:, -3:] = status[:, 1:].flip(1)

    def attach(idx):
        j = jobs[idx]
        name, lin, wrapped = items[idx]
        lin.weight.data = j.W_all
        U, L = (j.rows_all, j.dm_full) if j.right else (j.dm_full, j.rows_all)
        sf_out = (j.sf2 if quantizer.slim_quant else j.sf2.to(torch.bfloat16)).reshape(1, 1)
        lin.scaling_factor = sf_out
        lin.lora_rank = torch.tensor(j.rank_)
        left, right = ops.lora_adapters(U, L, j.rank_, prescale=True)               # prune.py:296-301
        lin.lora_left = torch.nn.Parameter(left)
        lin.lora_right = torch.nn.Parameter(right)
        lin.lora_fit_info = infos[idx]
        j.left, j.right_t, j.sf_out = left, right, sf_out
    fan_out(attach)
    mark("gather+attach")
    if trace:
        torch.cuda.synchronize()
        for (_, e0, h0), (name, e1, h1) in zip(marks, marks[1:]):
            print(f"[shard] {name}: gpu {e0.elapsed_time(e1):.2f} ms, host reached it at {h1:.2f} ms", flush=True)
    for idx, (name, lin, wrapped) in enumerate(items):
        quantizer.scaling_factor = lin.scaling_factor
        lin.register_forward_hook(add_lora_hook)
        if callback is not None:
            callback(name, lin, wrapped)
    return status
