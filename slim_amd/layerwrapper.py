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

    def __init__(self, layer, layer_id=0, layer_name="none", track_inp_sum=False, defer=False):
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
