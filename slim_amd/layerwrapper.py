"""Calibration statistics per linear layer (mirror of reference slim/layerwrapper.py:7-41)."""
import torch
import torch.nn as nn

from . import ops


class WrappedGPT:
    """Running per-input-column activation statistic of one nn.Linear.

    Same attributes as the reference wrapper (layerwrapper.py:12-23).  `scaler_row` is updated by
    the HIP column-sum-of-squares kernel; `inp_sum` (only read by an out-of-scope method of the
    reference) is kept when `track_inp_sum=True`.
    """

    def __init__(self, layer, layer_id=0, layer_name="none", track_inp_sum=False):
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
        if per_sample:
            ops.scaler_row_accum(flat, self.scaler_row, self.nsamples, n_chunks=b, batch_per_chunk=1)
        else:
            ops.scaler_row_accum(flat, self.scaler_row, self.nsamples, n_chunks=1, batch_per_chunk=b)
        self.nsamples += b
