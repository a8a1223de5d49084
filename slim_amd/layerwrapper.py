"""Calibration statistics per linear layer (mirror of reference slim/layerwrapper.py:7-41)."""
import torch
import torch.nn as nn

from . import ops


import contextlib

# Linears fed by the same tensor (q/k/v, gate/up of a decoder block) need its column sums once.  Inside a
# `shared_inputs()` scope -- the driver opens one around each block forward -- the sums are cached by the
# identity of the input tensor; the cache keeps the tensor alive, so its address cannot be reused by another
# tensor while the entry exists, and dies with the scope.
_shared = None


@contextlib.contextmanager
def shared_inputs():
    global _shared
    outer, _shared = _shared, {}
    try:
        yield
    finally:
        _shared = outer


def _sample_sums(flat, b):
    if _shared is None:
        return ops.colsumsq(flat, n_chunks=b)
    key = (flat.data_ptr(), tuple(flat.shape), flat._version, b)
    hit = _shared.get(key)
    if hit is None:
        hit = (flat, ops.colsumsq(flat, n_chunks=b))
        _shared[key] = hit
    return hit[1]


class WrappedGPT:
    """Running per-input-column activation statistic of one nn.Linear.

    Same attributes as the reference wrapper (layerwrapper.py:12-23).  `scaler_row` is updated by
    the HIP column-sum-of-squares kernel; `inp_sum` (only read by an out-of-scope method of the
    reference) is kept when `track_inp_sum=True`.
    """

    def __init__(self, layer, layer_id=0, layer_name="none", track_inp_sum=False, defer=False, collect_gram=False):
        self.layer = layer
        self.dev = self.layer.weight.device
        self.rows = layer.weight.data.shape[0]
        self.columns = layer.weight.data.shape[1]
        self.inp_sum = None
        self.inp_num = 0
        self.scaler_row = torch.zeros((self.columns), device=self.dev)
        self.nsamples = 0
        self.layer_id = layer_id
        self.layer_name = layer_name
        self.track_inp_sum = track_inp_sum
        # defer=True: keep the per-sample column sums instead of folding them into scaler_row; a run that
        # shards the samples over several GPUs gathers them into sample order first (finish_deferred)
        self.defer = defer
        self.sample_sums = []
        # collect_gram=True: also the Gram statistic of the reference's SparseGPT wrapper (sparsegpt.py:26, :28-39),
        # H = (2 / n) sum_samples X^T X in fp32 [K, K], the only place the reference forms a Gram matrix of the calibration
        # inputs; diag(H) / 2 equals scaler_row.  Each rank of a run that shards the calibration samples (tokens) holds the
        # H of its share until reduce_gram combines them.
        self.collect_gram = bool(collect_gram)
        self.H = torch.zeros((self.columns, self.columns), device=self.dev) if collect_gram else None
        self.gram_nsamples = 0

    def add_batch(self, inp, out=None, per_sample=False):
        """One forward call's input `inp` ([B, S, K] or [S, K], bf16).

        per_sample=False reproduces layerwrapper.py:25-41 (one update with batch = B).
        per_sample=True applies B successive batch-1 updates, i.e. what the reference's driver does
        when it feeds the samples one at a time; it lets a batched block forward keep that order.
        """
        if len(inp.shape) == 2:
            inp = inp.unsqueeze(0)
        b = inp.shape[0]
        if isinstance(self.layer, nn.Linear) and len(inp.shape) == 3:
            flat = inp.reshape((-1, inp.shape[-1]))
        else:
            flat = inp
        if self.track_inp_sum:
            t = flat.reshape(b, -1, flat.shape[-1]).sum(0) if per_sample else flat
            self.inp_sum = t.clone() if self.inp_sum is None else self.inp_sum + t
        self.inp_num += b if per_sample else 1
        if flat.dtype != torch.bfloat16:
            raise TypeError("WrappedGPT.add_batch expects bf16 activations")
        if self.collect_gram:
            # sparsegpt.py:36-39 with batch = b: H <- H n / (n + b) + (2 / (n + b)) X^T X.  The recurrence keeps
            # H = (2 / n) sum X^T X whatever the batching, so b single-sample updates give the same matrix up to the order of
            # fp32 additions (1e-6 relative); one update per call is taken for both settings of per_sample.
            ops.gram_accum(flat, self.H, self.gram_nsamples, b)
            self.gram_nsamples += b
        if self.defer:
            if not per_sample:
                raise ValueError("deferred statistics are kept per sample")
            self.sample_sums.append(_sample_sums(flat, b))
        elif per_sample and _shared is not None:
            # column sums (possibly shared with the other linears reading this tensor) + the recurrence:
            # the same arithmetic, in the same order, as the fused call below
            ops.scaler_row_fold(self.scaler_row, _sample_sums(flat, b), self.nsamples, batch_per_chunk=1)
        elif per_sample:
            ops.scaler_row_accum(flat, self.scaler_row, self.nsamples, n_chunks=b, batch_per_chunk=1)
        else:
            ops.scaler_row_accum(flat, self.scaler_row, self.nsamples, n_chunks=1, batch_per_chunk=b)
        self.nsamples += b

    def reduce_gram(self, all_reduce=None):
        """Token-sharded calibration: every rank has folded ITS samples into H = (2 / n_rank) sum X^T X.  The matrix over all
        samples is the mean of the ranks' matrices weighted by their sample counts, H = sum_ranks (n_rank / n) H_rank with
        n = sum n_rank: one all-reduce(SUM) of the pre-weighted matrices (and one of the counts).  all_reduce: in-place sum
        over the ranks (parallel.all_reduce); None: a single rank, nothing to do.  Afterwards H and gram_nsamples are the
        same on every rank and equal to a single-rank pass over all samples up to fp32 summation order."""
        if not self.collect_gram:
            raise ValueError("this wrapper does not collect the Gram matrix (collect_gram=True)")
        if all_reduce is None:
            return self.H
        count = torch.tensor([float(self.gram_nsamples)], dtype=torch.float64, device=self.H.device)
        local = float(count.item())
        all_reduce(count)
        total = float(count.item())
        if total > 0:
            self.H *= local / total
        all_reduce(self.H)
        self.gram_nsamples = int(total)
        return self.H

    def finish_deferred(self, gather=None):
        """Fold the kept per-sample sums (after `gather`, e.g. parallel.gather_sample_stats, has put the sums
        of all ranks into sample order) into scaler_row with the reference's recurrence."""
        sums = torch.cat(self.sample_sums, 0)
        self.sample_sums = []
        if gather is not None:
            sums = gather(sums)
        ops.scaler_row_fold(self.scaler_row, sums, 0, batch_per_chunk=1)
        self.nsamples = sums.shape[0]
        self.inp_num = sums.shape[0]
